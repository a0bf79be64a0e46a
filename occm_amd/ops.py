"""Thin torch-tensor wrappers over the C ABI of libocc_hip.so (pointers + sizes in, nothing else).

torch is used for device memory and streams only; every function here enqueues hand-written HIP
kernels on the current stream and raises ``OccError`` on failure.  There is no fallback.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_GELU_GRAD, ACT_GELU_KEEP_GRAD, ACT_MUL_AUX, ACT_NONE, ACT_RELU, ACT_SELU, ACT_TANH, OCC_AF32_WBF16, OCC_BF16, OCC_F32, OCC_F32_AS_BF16, OCC_F64,  # noqa: F401
                   GemmDesc, RowMap, check, dtype_code, lib, ptr, stream_ptr)


def _dev(t):
    if not t.is_cuda:
        raise _lib.OccError("occm_amd ops need CUDA/HIP tensors (got a %s tensor); there is no CPU path" % t.device)
    return t


# bench.py sets this to a list to time individual launches with events on the launch stream: (kind, start, end)
PROFILE = None


def rowmap(rows_per_batch, batch_stride, row_stride, rows_per_line=0, line_stride=0):
    return RowMap(int(rows_per_batch), int(batch_stride), int(row_stride), int(rows_per_line), int(line_stride))


def gemm_raw(M, N, K, A, a_map, W, ldw, C, c_map, c_dtype, ab_dtype, bias=None, act=ACT_NONE, alpha=1.0,
             R=None, r_map=None, r_dtype=OCC_F32, a_seg=None, groups=None, aux=None, a_dequant=None, w_dequant=None, c_f8=None, c_colsum=None):
    """Direct descriptor-level call.  A/W/C/R/bias are ints (device addresses) or tensors."""
    d = GemmDesc()
    d.M, d.N, d.K = int(M), int(N), int(K)
    d.A = A if isinstance(A, int) else A.data_ptr()
    d.a_map = a_map
    if a_seg is not None:
        d.a_nseg, d.a_seg_len, d.a_seg_stride = [int(v) for v in a_seg]
    else:
        d.a_nseg, d.a_seg_len, d.a_seg_stride = 1, int(K), 0
    d.W = W if isinstance(W, int) else W.data_ptr()
    d.ldw = int(ldw)
    d.bias = None if bias is None else (bias if isinstance(bias, int) else bias.data_ptr())
    if R is not None:
        d.R = R if isinstance(R, int) else R.data_ptr()
        d.r_map = r_map
        d.r_dtype = r_dtype
    d.C = C if isinstance(C, int) else C.data_ptr()
    d.c_map = c_map
    d.c_dtype = c_dtype
    d.ab_dtype = ab_dtype
    d.act = act
    d.alpha = float(alpha)
    if groups is not None:
        d.n_groups, d.a_group_stride, d.w_group_stride, d.c_group_stride = [int(v) for v in groups]
    if aux is not None:
        d.aux = aux if isinstance(aux, int) else aux.data_ptr()
    if a_dequant is not None:
        d.a_dequant = a_dequant if isinstance(a_dequant, int) else a_dequant.data_ptr()
    if w_dequant is not None:
        d.w_dequant = w_dequant if isinstance(w_dequant, int) else w_dequant.data_ptr()
    if c_f8 is not None:                       # (u8 buffer, scale scalar, amax scalar, fp8 format): fp8 copy of the bf16 result from the same epilogue
        d.c_f8, d.c_f8_scale, d.c_f8_amax, d.c_f8_fmt = c_f8[0].data_ptr(), c_f8[1].data_ptr(), c_f8[2].data_ptr(), int(c_f8[3])
    if c_colsum is not None:                   # f32 [N] += column sums of the bf16 result (bias gradient), from the same epilogue
        if isinstance(c_colsum, tuple):        # (out, site scratch): partial sums only, the caller runs FinalizeBatch after the backward pass
            c_colsum, ws = c_colsum
            d.c_colsum_defer = 1
        else:
            ws = small_scratch(2 * ((int(M) + 207) // 208) * int(N))          # one partial row per wave row of a tile; 208 rows = the smallest tile (occ_gemm checks the size)
        d.c_colsum, d.c_colsum_ws, d.c_colsum_ws_floats = c_colsum.data_ptr(), ws.data_ptr(), ws.numel()
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().occ_gemm(ctypes.byref(d), stream_ptr()), "occ_gemm")
        e1.record()
        PROFILE.append(("gemm_bf16" if ab_dtype in (OCC_BF16, OCC_AF32_WBF16, _lib.OCC_FP8_E4M3, _lib.OCC_FP8_E5M2) else "gemm_f32", e0, e1))
        return
    check(lib().occ_gemm(ctypes.byref(d), stream_ptr()), "occ_gemm")


def linear(x, w, bias=None, act=ACT_NONE, residual=None, out=None, out_dtype=None, alpha=1.0, ab_dtype=None):
    """y = act(alpha * x @ w.T + bias) + residual for 2-D row-major x [M,K], w [N,K].  ab_dtype: the GEMM's arithmetic when it is not the
    operands' storage type (OCC_F32X3 on f32 operands)."""
    _dev(x); _dev(w)
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and x.dtype == w.dtype and x.is_contiguous() and w.is_contiguous()
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=out_dtype or x.dtype)
    full = rowmap(M, 0, K)
    gemm_raw(M, N, K, x, full, w, K, out, rowmap(M, 0, N), dtype_code(out), dtype_code(x) if ab_dtype is None else ab_dtype, bias=bias, act=act, alpha=alpha,
             R=residual, r_map=None if residual is None else rowmap(M, 0, N),
             r_dtype=OCC_F32 if residual is None else dtype_code(residual))
    return out


def layernorm(x, gamma, beta, eps=1e-5, gelu=False, out=None, out_dtype=None):
    _dev(x)
    C = x.shape[-1]
    rows = x.numel() // C
    assert x.is_contiguous()
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=out_dtype or x.dtype)
    check(lib().occ_layernorm(ptr(x), dtype_code(x), ptr(out), dtype_code(out), ptr(gamma), ptr(beta), rows, C, eps, int(gelu), stream_ptr()),
          "occ_layernorm")
    return out


def conv0_ln_gelu(wav, w, bias, gamma, beta, k, stride, out_dtype, eps=1e-5, out=None):
    _dev(wav)
    B, L = wav.shape
    Tout = (L - k) // stride + 1
    if out is None:
        out = torch.empty(B, Tout, 512, device=wav.device, dtype=out_dtype)
    check(lib().occ_conv0_ln_gelu(ptr(wav), ptr(w), ptr(bias), ptr(gamma), ptr(beta), ptr(out), dtype_code(out), B, L, Tout, 512, k, stride, eps,
                                  stream_ptr()), "occ_conv0_ln_gelu")
    return out


def attention(qkv, B, T, H, hd, scale, out=None, lse=None):
    """lse: optional f32 [B*H, T] output (log2-sum-exp2 of the scaled scores) kept for attention_bwd."""
    _dev(qkv)
    D = H * hd
    assert qkv.shape == (B * T, 3 * D) and qkv.is_contiguous()
    if out is None:
        out = torch.empty(B * T, D, device=qkv.device, dtype=qkv.dtype)
    check(lib().occ_attention(ptr(qkv), ptr(out), dtype_code(qkv), B, T, H, hd, 3 * D, D, float(scale), ptr(lse), stream_ptr()), "occ_attention")
    return out


def split3_bf16(x, out=None, mode=0, x_map=None, rows=None, K=None):
    """f32 [rows, K] -> bf16 [rows, 3K]: [hi | lo | hi] (mode 0, activations) or [hi | hi | lo] (mode 1, weights); the two then multiply
    as ONE bf16 GEMM of depth 3K that yields xh.wh + xl.wh + xh.wl (occ_split3_bf16)."""
    if rows is None:
        rows, K = x.shape
    if out is None:
        out = torch.empty(rows, 3 * K, device=x.device, dtype=torch.bfloat16)
    m = x_map if x_map is not None else rowmap(rows, 0, K)
    check(lib().occ_split3_bf16(x if isinstance(x, int) else ptr(x), ctypes.byref(m), ptr(out), int(rows), int(K), int(mode), stream_ptr()), "occ_split3_bf16")
    return out


def attention_varlen(qkv, B, T, H, hd, scale, kv_len, out=None):
    """Attention over a zero-padded batch: kv_len int32 [B] (device) = valid frames per utterance; keys at or past it are masked, so the
    first kv_len[b] rows of utterance b equal its un-padded single-utterance result (f32 arithmetic, storage f32 or bf16)."""
    _dev(qkv)
    D = H * hd
    assert qkv.shape == (B * T, 3 * D) and qkv.is_contiguous() and kv_len.dtype == torch.int32 and kv_len.numel() == B
    if out is None:
        out = torch.empty(B * T, D, device=qkv.device, dtype=qkv.dtype)
    check(lib().occ_attention_varlen(ptr(qkv), ptr(out), dtype_code(qkv), B, T, H, hd, 3 * D, D, float(scale), ptr(kv_len), stream_ptr()), "occ_attention_varlen")
    return out


def attention_bwd(qkv, o, dout, lse, B, T, H, hd, scale, dqkv=None):
    D = H * hd
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    acc = torch.empty(B * T, D, device=qkv.device, dtype=torch.float32) if T > 256 else None      # the key blocks of a head meet here
    check(lib().occ_attention_bwd(ptr(qkv), ptr(o), ptr(dout), ptr(lse), ptr(dqkv), B, T, H, hd, 3 * D, D, float(scale), ptr(acc), stream_ptr()),
          "occ_attention_bwd")
    return dqkv


def attention_bwd_bias(qkv, o, dout, lse, B, T, H, hd, scale, dbias, dqkv=None, defer=None):
    """attention_bwd (T <= 256) that also adds the column sums of dqkv -- the q|k|v bias gradient -- into dbias f32 [3D].
    defer: this call site's own record buffer (>= B*H*3*hd floats): the per-(batch, head) sums stay there, no finalize launch (FinalizeBatch)."""
    D = H * hd
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    ws = defer if defer is not None else small_scratch()
    check(lib().occ_attention_bwd_bias(ptr(qkv), ptr(o), ptr(dout), ptr(lse), ptr(dqkv), B, T, H, hd, 3 * D, D, float(scale), ptr(dbias), ptr(ws), ws.numel(),
                                       1 if defer is not None else 0, stream_ptr()), "occ_attention_bwd_bias")
    return dqkv


def attention_dropout(qkv, B, T, H, hd, scale, keep, p, out=None, lse=None):
    """attention with fairseq's attention_dropout: keep u8 [B*H, T, Tp] (Tp = T rounded up to 4), p the drop probability."""
    _dev(qkv)
    D, Tp = H * hd, (T + 3) // 4 * 4
    assert qkv.shape == (B * T, 3 * D) and qkv.is_contiguous() and qkv.dtype == torch.bfloat16
    assert keep.dtype == torch.uint8 and keep.numel() == B * H * T * Tp and keep.is_contiguous()
    if out is None:
        out = torch.empty(B * T, D, device=qkv.device, dtype=qkv.dtype)
    check(lib().occ_attention_dropout(ptr(qkv), ptr(out), B, T, H, hd, 3 * D, D, float(scale), ptr(lse), ptr(keep), float(p), stream_ptr()), "occ_attention_dropout")
    return out


def attention_bwd_dropout(qkv, o, dout, lse, B, T, H, hd, scale, keep, p, dqkv=None):
    D = H * hd
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    acc = torch.empty(B * T, D, device=qkv.device, dtype=torch.float32) if T > 256 else None
    check(lib().occ_attention_bwd_dropout(ptr(qkv), ptr(o), ptr(dout), ptr(lse), ptr(dqkv), B, T, H, hd, 3 * D, D, float(scale), ptr(acc), ptr(keep), float(p),
                                          stream_ptr()), "occ_attention_bwd_dropout")
    return dqkv


def dropout_mask(mask, p, seed=0, stream_id=0):
    """Fills the u8 tensor `mask` with the keep-mask occ_dropout_ex(generate=1) would draw for (numel, p, seed, stream_id)."""
    check(lib().occ_dropout_mask(ptr(mask), mask.numel(), float(p), int(seed), int(stream_id), stream_ptr()), "occ_dropout_mask")
    return mask


def _p(t):
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t) if isinstance(t, int) else ctypes.c_void_p(t.data_ptr())


def transpose_bf16(src, dst, rows, cols, ld_src=None, ld_dst=None, colsum=None, src_dtype=None):
    """dst[c, r] = bf16(src[r, c]); dst is a bf16 [cols, ld_dst] buffer whose pad columns the caller keeps at zero.
    colsum (optional f32 [cols]) accumulates the column sums of src (bias gradient).  src / dst may be int addresses
    (then src_dtype and ld_dst are required)."""
    sd = src_dtype if src_dtype is not None else dtype_code(src)
    check(lib().occ_transpose_bf16(_p(src), sd, _p(dst), int(rows), int(cols), int(ld_src or cols), int(ld_dst or dst.shape[-1]),
                                   _p(colsum), stream_ptr()), "occ_transpose_bf16")
    return dst


def transpose_bf16_rows(src, src_map, dst, rows, cols, ld_dst, colsum=None, src_dtype=None):
    """Like transpose_bf16 with the source rows addressed through a row map."""
    sd = src_dtype if src_dtype is not None else dtype_code(src)
    check(lib().occ_transpose_bf16_rows(_p(src), sd, ctypes.byref(src_map), _p(dst), int(rows), int(cols), int(ld_dst), _p(colsum), stream_ptr()),
          "occ_transpose_bf16_rows")
    return dst


class Fp8Batch:
    """A fixed set of bf16 tensors quantised (or only measured: dst None) in ONE launch (occ_fp8_quantize_batch).  scale / amax are
    one-element device tensors (views into the per-site arrays) or None; the job table is uploaded on the first run()."""

    def __init__(self, fmt):
        self.fmt, self.jobs, self.chunks, self._dev, self._keep = int(fmt), [], 0, None, []

    def add(self, src, dst=None, scale=None, amax=None):
        n = src.numel()
        assert src.dtype == torch.bfloat16 and n % 8 == 0 and src.is_contiguous()
        self.jobs.append((src.data_ptr(), 0 if dst is None else dst.data_ptr(), n, 0 if scale is None else scale.data_ptr(),
                          0 if amax is None else amax.data_ptr(), self.chunks))
        self.chunks += (n + 8191) // 8192
        self._keep.append((src, dst, scale, amax))
        self._dev = None

    def run(self):
        if not self.jobs:
            return
        if self._dev is None:
            import numpy as np
            self._dev = torch.from_numpy(np.asarray(self.jobs, dtype=np.int64)).cuda()
        check(lib().occ_fp8_quantize_batch(ptr(self._dev), len(self.jobs), self.chunks, self.fmt, stream_ptr()), "occ_fp8_quantize_batch")


class TransposeBatch:
    """A fixed set of transposes dst[c, r] = bf16(src[r, c]) run as ONE launch (occ_transpose_bf16_batch).  add() takes tensors or int device
    addresses; the job table is uploaded on the first run() and the operands must not move afterwards."""

    def __init__(self):
        self.jobs, self.tiles, self._dev = [], 0, None

    def add(self, src, dst, rows, cols, ld_src=None, ld_dst=None, src_dtype=None):
        sd = src_dtype if src_dtype is not None else dtype_code(src)
        sp = src if isinstance(src, int) else src.data_ptr()
        dp = dst if isinstance(dst, int) else dst.data_ptr()
        self.jobs.append((sp, dp, int(rows), int(cols), int(ld_src or cols), int(ld_dst or dst.shape[-1]), self.tiles, int(sd)))
        self.tiles += ((int(rows) + 63) // 64) * ((int(cols) + 63) // 64)
        self._dev = None

    def run(self):
        if not self.jobs:
            return
        if self._dev is None:
            import numpy as np
            self._dev = torch.from_numpy(np.asarray(self.jobs, dtype=np.int64)).cuda()
        check(lib().occ_transpose_bf16_batch(ptr(self._dev), len(self.jobs), self.tiles, stream_ptr()), "occ_transpose_bf16_batch")


FP8_MAX = {_lib.OCC_FP8_E4M3: 448.0, _lib.OCC_FP8_E5M2: 57344.0}


def fp8_quantize(src, dst, fmt, scale=None, amax=None):
    """dst (uint8, src.numel()) = fp8(src * scale); amax (device scalar) is raised to max |src|."""
    check(lib().occ_fp8_quantize(ptr(src), dtype_code(src), ptr(dst), int(fmt), src.numel(), ptr(scale), ptr(amax), stream_ptr()), "occ_fp8_quantize")
    return dst


def fp8_amax(src, amax):
    check(lib().occ_fp8_amax(ptr(src), dtype_code(src), src.numel(), ptr(amax), stream_ptr()), "occ_fp8_amax")


def fp8_update_scales(amax, scale, inv_scale, fmt, margin=1.0):
    check(lib().occ_fp8_update_scales(ptr(amax), ptr(scale), ptr(inv_scale), amax.numel(), FP8_MAX[fmt], float(margin), stream_ptr()), "occ_fp8_update_scales")


def dropout_ex(x, y, mask, p, seed=0, stream_id=0, generate=False, residual=None, scale=1.0):
    """y = residual + scale * x * keep / (1 - p) (x, y f32 or bf16, may alias; mask u8 or None)."""
    check(lib().occ_dropout_ex(ptr(x), dtype_code(x), ptr(y), dtype_code(y), ptr(mask), ptr(residual), x.numel(), float(p), float(scale), int(seed),
                               int(stream_id), int(generate), stream_ptr()), "occ_dropout_ex")
    return y


_SCRATCH, _SCRATCH_RETIRED = {}, []


def small_scratch(nfloats=512 * 2048):
    """Per (device, stream) f32 scratch of AT LEAST ``nfloats`` for kernels that hand partial sums to a second launch (calls sharing it are
    ordered on that stream).  It only grows; a replaced buffer stays allocated because a captured HIP graph may still hold its address."""
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    t = _SCRATCH.get(key)
    if t is None or t.numel() < nfloats:
        if t is not None:
            _SCRATCH_RETIRED.append(t)
        t = _SCRATCH[key] = torch.empty(max(int(nfloats), 512 * 2048), device="cuda", dtype=torch.float32)
    return t


def layernorm_bwd_ex(dy, x, gamma, beta, dres, dx, dx_bf16, dx_bf16_map, dgamma, dbeta, gelu, eps=1e-5):
    """General LayerNorm backward (x f32 or bf16, optional fused GELU', f32 and/or row-mapped bf16 outputs)."""
    C = x.shape[-1]
    rows = x.numel() // C
    sc = small_scratch(256 * 2 * C)          # at most 256 workgroups leave partial sums (frontend_bwd.hip lnb_blocks)
    check(lib().occ_layernorm_bwd_ex(_p(dy), dtype_code(dy), _p(x), dtype_code(x), _p(gamma), _p(beta), _p(dres), _p(dx), _p(dx_bf16),
                                     ctypes.byref(dx_bf16_map) if dx_bf16_map is not None else None, _p(dgamma), _p(dbeta), rows, C, float(eps),
                                     int(gelu), ptr(sc), sc.numel(), stream_ptr()), "occ_layernorm_bwd_ex")


def layernorm_bwd(dy, x, gamma, dres, dx, dgamma, dbeta, eps=1e-5, dx_bf16=None):
    C = x.shape[-1]
    rows = x.numel() // C
    sc = small_scratch(256 * 2 * C)          # at most 256 workgroups leave partial sums (frontend_bwd.hip lnb_blocks)
    check(lib().occ_layernorm_bwd(ptr(dy), dtype_code(dy), ptr(x), ptr(gamma), ptr(dres), ptr(dx), ptr(dx_bf16), ptr(dgamma), ptr(dbeta), rows, C,
                                  float(eps), ptr(sc), sc.numel(), stream_ptr()), "occ_layernorm_bwd")
    return dx


def layernorm_fp8(x, gamma, beta, out, out_f8, f8_scale, f8_amax, eps=1e-5):
    """LayerNorm f32 -> bf16 `out` plus the e4m3 copy `out_f8` (u8, same shape) at the delayed scale, recording |max| in f8_amax."""
    C = x.shape[-1]
    check(lib().occ_layernorm_fp8(ptr(x), ptr(out), ptr(out_f8), ptr(f8_scale), ptr(f8_amax), ptr(gamma), ptr(beta), x.numel() // C, C, float(eps), stream_ptr()),
          "occ_layernorm_fp8")
    return out


def layernorm_bwd_fused(dy, x, gamma, dres, dx, dgamma, dbeta, dx_bf16, dbias=None, dx_f8=None, f8_scale=None, f8_amax=None, eps=1e-5, defer=None):
    """layernorm_bwd with the bias gradient of the preceding Linear (column sums of dx) and / or the e5m2 copy of the bf16 dx folded in.
    defer: a scratch tensor of this call site's own (>= 768 * C floats): the partial sums stay there, no finalize launch (FinalizeBatch)."""
    C = x.shape[-1]
    rows = x.numel() // C
    sc = defer if defer is not None else small_scratch(256 * 3 * C)
    check(lib().occ_layernorm_bwd_fused(ptr(dy), dtype_code(dy), ptr(x), ptr(gamma), ptr(dres), ptr(dx), ptr(dx_bf16), ptr(dgamma), ptr(dbeta), ptr(dbias),
                                        ptr(dx_f8), ptr(f8_scale), ptr(f8_amax), rows, C, float(eps), ptr(sc), sc.numel(), 1 if defer is not None else 0, stream_ptr()),
          "occ_layernorm_bwd_fused")
    return dx


class FinalizeBatch:
    """Job table for occ_finalize_batch: the partial sums that deferred producers (layernorm_bwd_fused(defer=), gemm_raw(c_colsum=(out, ws)),
    attention_bwd_bias(defer=)) left in their per-site buffers are added into the gradients with ONE launch.  The table lives on the
    device and is re-uploaded only when the list of jobs changes (it does not between steps of one shape)."""

    def __init__(self):
        self.jobs, self._key, self._dev, self._blocks = [], None, None, 0
        self._refs = {}                        # data_ptr -> tensor: the table holds raw addresses, so it keeps their owners alive

    def begin(self):
        self.jobs = []

    def _hold(self, *tensors):
        for t in tensors:
            if t is not None:
                self._refs[t.data_ptr()] = t

    def add_rows(self, partials, n_rows, out0, out1=None, out2=None, C=None):
        """kind 0: out[i] += sum over n_rows rows of partials [n_rows, tot]; i < C -> out0, < 2C -> out1, else out2 (tot = C * number of outs given).
        For gemm_raw(c_colsum=(out, ws)) pass n_rows = 2 * ceil(M / 208) and a ws that started as ZEROS: occ_gemm writes 2 * ceil(M / tile rows)
        partial rows with 208-, 224- or 256-row tiles (its own choice per shape); the rows it does not write must read as zero."""
        C = int(C if C is not None else out0.numel())
        tot = C * (1 + (out1 is not None) + (out2 is not None))
        if out2 is not None and out1 is None:
            raise ValueError("out2 needs out1")
        self._hold(partials, out0, out1, out2)
        self.jobs.append((0, partials.data_ptr(), out0.data_ptr(), 0 if out1 is None else out1.data_ptr(), 0 if out2 is None else out2.data_ptr(), int(n_rows), tot, C,
                          (tot + 63) // 64))

    def add_ln(self, partials, rows, C, dgamma, dbeta, dbias):
        """The fused LayerNorm backward's [block][3][C] partial sums (dbias may be None: the third set is then skipped by the sum)."""
        n = min(256, (int(rows) + 47) // 48)
        self._hold(partials, dgamma, dbeta, dbias)
        self.jobs.append((0, partials.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), 0 if dbias is None else dbias.data_ptr(), n, 3 * int(C), int(C), (3 * int(C) + 63) // 64))

    def add_attention_bias(self, records, B, H, hd, dbias):
        self._hold(records, dbias)
        self.jobs.append((1, records.data_ptr(), dbias.data_ptr(), 0, 0, int(B), int(H), int(hd), 3 * int(H)))

    def run(self):
        if not self.jobs:
            return
        key = tuple(self.jobs)
        if key != self._key:
            arr = (_lib.FinalizeJob * len(self.jobs))()
            first = 0
            for j, (kind, part, o0, o1, o2, n0, n1, n2, nb) in zip(arr, self.jobs):
                j.partials, j.out0, j.out1, j.out2 = part, o0, o1 or None, o2 or None
                j.kind, j.n0, j.n1, j.n2, j.first_block, j.n_blocks = kind, n0, n1, n2, first, nb
                first += nb
            raw = bytes(arr)
            self._dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
            self._key, self._blocks, self._n = key, first, len(self.jobs)
        check(lib().occ_finalize_batch(self._dev.data_ptr(), self._n, self._blocks, stream_ptr()), "occ_finalize_batch")


# ---------------------------------------------------------------------------------- RawBoost ---
def rawboost_fir_bank(x, coef, ntaps, powers):
    """x f32/f64 [B,L]; coef f64 [B,F,max_taps]; ntaps i32 [B,F] -> y f64 [B,L]."""
    _dev(x)
    B, L = x.shape
    F, max_taps = coef.shape[1], coef.shape[2]
    y = torch.empty(B, L, device=x.device, dtype=torch.float64)
    check(lib().occ_rawboost_fir_bank(ptr(x), dtype_code(x), ptr(y), ptr(coef), ptr(ntaps), B, L, F, max_taps, int(powers), stream_ptr()),
          "occ_rawboost_fir_bank")
    return y


def _partials(B, L, device):
    return torch.empty(B, (L + 4095) // 4096, 4, device=device, dtype=torch.float64)


def rawboost_center_norm(y, subtract_mean, norm_mode):
    B, L = y.shape
    check(lib().occ_rawboost_center_norm(ptr(y), B, L, int(subtract_mean), int(norm_mode), ptr(_partials(B, L, y.device)), stream_ptr()),
          "occ_rawboost_center_norm")
    return y


def rawboost_isd_scatter(y, pos, fr, n, g_sd):
    B, L = y.shape
    check(lib().occ_rawboost_isd_scatter(ptr(y), ptr(pos), ptr(fr), ptr(n), B, L, pos.shape[1], float(g_sd), stream_ptr()),
          "occ_rawboost_isd_scatter")
    return y


def rawboost_ssi_mix(x, noise, snr):
    B, L = x.shape
    out = torch.empty_like(x)
    check(lib().occ_rawboost_ssi_mix(ptr(x), ptr(noise), ptr(snr), ptr(out), B, L, ptr(_partials(B, L, x.device)), stream_ptr()),
          "occ_rawboost_ssi_mix")
    return out


def add_f64(a, b):
    out = torch.empty_like(a)
    check(lib().occ_add_f64(ptr(a), ptr(b), ptr(out), a.numel(), stream_ptr()), "occ_add_f64")
    return out


def cast(src, dtype):
    dst = torch.empty(src.shape, device=src.device, dtype=dtype)
    check(lib().occ_cast(ptr(src), dtype_code(src), ptr(dst), dtype_code(dst), src.numel(), stream_ptr()), "occ_cast")
    return dst


def philox_fill(shape, dtype, seed, stream_id, normal, device="cuda"):
    dst = torch.empty(shape, device=device, dtype=dtype)
    check(lib().occ_philox_fill(ptr(dst), dtype_code(dst), dst.numel(), int(seed), int(stream_id), int(normal), stream_ptr()), "occ_philox_fill")
    return dst


def notch_coeffs_host(bands, G, fs, max_taps=512):
    """Host-only: (fc,bw,c) list + gain -> (coef f64[max_taps], ntaps)."""
    fc = np.ascontiguousarray([b[0] for b in bands], dtype=np.float64)
    bw = np.ascontiguousarray([b[1] for b in bands], dtype=np.float64)
    c = np.ascontiguousarray([b[2] for b in bands], dtype=np.int32)
    out = np.zeros(max_taps, dtype=np.float64)
    nt = np.zeros(1, dtype=np.int32)
    check(lib().occ_notch_coeffs_host(ptr(fc), ptr(bw), ptr(c), len(bands), float(G), float(fs), ptr(out), ptr(nt), max_taps),
          "occ_notch_coeffs_host")
    return out, int(nt[0])


# ------------------------------------------------------------------------------------ losses ---
def compactness_loss(emb, n_groups=1, group=None, scale=1.0, want_grad=False):
    _dev(emb)
    rows, E = emb.shape
    group = group or rows // n_groups
    loss = torch.empty(1, device=emb.device, dtype=torch.float32)
    demb = torch.empty_like(emb) if want_grad else None
    check(lib().occ_compactness_loss(ptr(emb), ptr(loss), ptr(demb), n_groups, group, E, float(scale), stream_ptr()), "occ_compactness_loss")
    return loss, demb


def ce_loss(logits, labels, scale=1.0, want_grad=False):
    _dev(logits)
    B, C = logits.shape
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    dl = torch.empty_like(logits) if want_grad else None
    check(lib().occ_ce_loss(ptr(logits), ptr(labels), ptr(loss), ptr(dl), B, C, float(scale), stream_ptr()), "occ_ce_loss")
    return loss, dl


def pair_dist_loss(emb, pairs, weights, bias=0.0, relu=False, scale=1.0, want_grad=False):
    """loss [1] = act(bias + sum_k weights[k] * ||emb[i_k] - emb[j_k] + 1e-6||); pairs = [(i, j), ...] (custom_loss.py:32-74).
    As in every loss kernel of the library the returned VALUE is the unweighted loss and ``scale`` (the loss weight of the training step)
    multiplies the GRADIENT only: demb = d(scale * loss) / d emb."""
    _dev(emb)
    R, E = emb.shape
    if any(not (0 <= i < R and 0 <= j < R) for i, j in pairs):
        raise _lib.OccError("pair_dist_loss: a pair names a row outside the %d embeddings" % R)
    pi = torch.tensor([i for i, _ in pairs], dtype=torch.int32, device=emb.device)
    pj = torch.tensor([j for _, j in pairs], dtype=torch.int32, device=emb.device)
    w = torch.tensor(list(weights), dtype=torch.float32, device=emb.device)
    loss = torch.empty(1, device=emb.device, dtype=torch.float32)
    demb = torch.empty_like(emb) if want_grad else None
    check(lib().occ_pair_dist_loss(ptr(emb), ptr(pi), ptr(pj), ptr(w), len(pairs), float(bias), int(bool(relu)), ptr(loss), ptr(demb), R, E, float(scale),
                                   stream_ptr()), "occ_pair_dist_loss")
    return loss, demb


def pairwise_dist(ref, emb):
    _dev(emb)
    N, E = emb.shape
    dist = torch.empty(N, device=emb.device, dtype=torch.float32)
    check(lib().occ_pairwise_dist(ptr(ref), ptr(emb), ptr(dist), N, E, stream_ptr()), "occ_pairwise_dist")
    return dist


class AdamMulti:
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8) semantics (oc_training.py:324) as one launch."""

    def __init__(self, params, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, bf16_copies=None):
        """bf16_copies: optional list aligned with params (entries may be None): bf16 tensors of the same numel that receive the updated values."""
        self.params = [p for p in params]
        self.bf16_copies = list(bf16_copies) if bf16_copies is not None else None
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        dev = self.params[0].device
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self._p = torch.tensor([p.data_ptr() for p in self.params], dtype=torch.int64, device=dev)
        self._m = torch.tensor([p.data_ptr() for p in self.exp_avg], dtype=torch.int64, device=dev)
        self._v = torch.tensor([p.data_ptr() for p in self.exp_avg_sq], dtype=torch.int64, device=dev)
        self._sizes = torch.tensor([p.numel() for p in self.params], dtype=torch.int64, device=dev)
        self._steps = torch.zeros(len(self.params), dtype=torch.int32, device=dev)
        self._b = None
        if self.bf16_copies is not None and any(b is not None for b in self.bf16_copies):
            assert all(b is None or (b.dtype == torch.bfloat16 and b.numel() == p.numel()) for b, p in zip(self.bf16_copies, self.params))
            self._b = torch.tensor([0 if b is None else b.data_ptr() for b in self.bf16_copies], dtype=torch.int64, device=dev)
        self._max = max(p.numel() for p in self.params)
        self._g_host = None
        self._g = None

    def _grad_table(self, grads):
        ptrs = [0 if g is None else g.data_ptr() for g in grads]
        if self._g_host != ptrs:
            self._g_host = ptrs
            self._g = torch.tensor(ptrs, dtype=torch.int64, device=self.params[0].device)

    def step(self, grads, grad_scale=1.0):
        """grads: list aligned with params (None = no gradient)."""
        self.set_grads(grads)
        self.step_range(0, len(self.params), grad_scale)

    def set_grads(self, grads):
        """Registers the gradient tensors of this optimizer step for step_range()."""
        self._grad_table(grads)
        self.step_count += 1

    def step_range(self, first, count, grad_scale=1.0):
        """The update of tensors [first, first + count) only, on the current stream: an overlapped trainer updates a transformer layer as
        soon as its gradients are final.  Every tensor must be stepped exactly once per optimizer step, after one set_grads()."""
        if count <= 0:
            return
        off8, off4 = first * 8, first * 4
        P = lambda t, o: ctypes.c_void_p(t.data_ptr() + o)
        mx = max(p.numel() for p in self.params[first:first + count])
        check(lib().occ_adam_multi(P(self._p, off8), P(self._g, off8), P(self._m, off8), P(self._v, off8), P(self._sizes, off8), P(self._steps, off4),
                                   count, mx, self.lr, self.betas[0], self.betas[1], self.eps, float(grad_scale),
                                   P(self._b, off8) if self._b is not None else None, stream_ptr()), "occ_adam_multi")
