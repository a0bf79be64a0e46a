"""Wrappers over the AASIST back-end entry points of libocc_hip.so (f32, channels-last).

Arguments named ``*`` that accept "addr" take either a tensor or an int device address (tensor
``data_ptr()`` plus a byte offset) so interiors of zero-padded buffers can be addressed directly.
"""
import ctypes

import numpy as np
import torch

from ._lib import GemmTnDesc, MasterDesc, MasterGrads, ReadoutDesc, ReadoutGrads, RowMap, check, lib, stream_ptr
from .ops import rowmap  # noqa: F401


def _a(t):
    if t is None:
        return ctypes.c_void_p(0)
    if isinstance(t, int):
        return ctypes.c_void_p(t)
    return ctypes.c_void_p(t.data_ptr())


def _ai(t):
    if t is None:
        return None
    return t if isinstance(t, int) else t.data_ptr()


def full(rows, C):
    return RowMap(int(rows), 0, int(C), 0, 0)


_TN_WS = {}
TN_WORKSPACE_BYTES = 256 * 262144 + 4096        # one 256 KiB slab per CU + the ticket block


def tn_workspace():
    """Device scratch of occ_gemm_tn's large bf16 products, one per (device, stream): calls sharing it must be ordered on that stream."""
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    if key not in _TN_WS:
        _TN_WS[key] = torch.empty(TN_WORKSPACE_BYTES, device="cuda", dtype=torch.uint8)
    return _TN_WS[key]


def gemm_tn(M, N1, N2, A, a_map, Bm, b_map, C, ldc, b_seg=None, alpha=1.0, colsum_out=None, a_bf16=False, b_bf16=False, bf16_mfma=False, groups=None, c_is_zero=False):
    """groups = (n_groups, a_stride, b_stride, c_stride) in elements: that many independent products in one launch (grouped conv)."""
    d = GemmTnDesc()
    d.c_is_zero = 1 if c_is_zero else 0
    if groups is not None:
        d.n_groups, d.a_group_stride, d.b_group_stride, d.c_group_stride = [int(x) for x in groups]
    if a_bf16 and b_bf16 and bf16_mfma and N1 % 256 == 0 and N2 % 256 == 0 and M >= 1024:
        ws = tn_workspace()
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    d.M, d.N1, d.N2 = int(M), int(N1), int(N2)
    d.A, d.a_map = _ai(A), a_map
    d.B, d.b_map = _ai(Bm), b_map
    if b_seg is not None:
        d.b_nseg, d.b_seg_len, d.b_seg_stride = [int(v) for v in b_seg]
    else:
        d.b_nseg, d.b_seg_len, d.b_seg_stride = 1, int(N2), 0
    d.C, d.ldc, d.alpha = _ai(C), int(ldc), float(alpha)
    d.colsum = _ai(colsum_out)
    d.a_dtype, d.b_dtype = int(bool(a_bf16)), int(bool(b_bf16))
    d.compute = 1 if bf16_mfma else 0          # OCC_BF16 / OCC_F32
    from . import ops
    if ops.PROFILE is not None:                # bench.py: time this launch with events on the launch stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().occ_gemm_tn(ctypes.byref(d), stream_ptr()), "occ_gemm_tn")
        e1.record()
        ops.PROFILE.append(("gemm_bf16" if (a_bf16 and b_bf16 and bf16_mfma) else "gemm_tn_other", e0, e1))
        return
    check(lib().occ_gemm_tn(ctypes.byref(d), stream_ptr()), "occ_gemm_tn")


def _tn_desc_bf16(M, N1, N2, A, a_map, Bm, b_map, C, ldc, colsum_out, ws):
    d = GemmTnDesc()
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    d.M, d.N1, d.N2 = int(M), int(N1), int(N2)
    d.A, d.a_map, d.B, d.b_map = _ai(A), a_map, _ai(Bm), b_map
    d.b_nseg, d.b_seg_len, d.b_seg_stride = 1, int(N2), 0
    d.C, d.ldc, d.alpha = _ai(C), int(ldc), 1.0
    d.colsum = _ai(colsum_out)
    d.a_dtype, d.b_dtype, d.compute = 1, 1, 1
    return d


def gemm_tn_pair(M, first, second, c_is_zero=False):
    """Two bf16 weight gradients with the same reduction rows in one launch where the library can (occ_gemm_tn_pair); first / second =
    (N1, N2, A, a_map, B, b_map, C, ldc, colsum_out).  c_is_zero: both C hold zeros (gradient buffers cleared since their last use) -- the
    slab reduce then stores instead of reading C back (occ_gemm_tn_desc.c_is_zero)."""
    ws = tn_workspace()
    d0, d1 = _tn_desc_bf16(M, *first, ws), _tn_desc_bf16(M, *second, ws)
    d0.c_is_zero = d1.c_is_zero = 1 if c_is_zero else 0
    from . import ops
    if ops.PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().occ_gemm_tn_pair(ctypes.byref(d0), ctypes.byref(d1), stream_ptr()), "occ_gemm_tn_pair")
        e1.record()
        ops.PROFILE.append(("gemm_bf16", e0, e1))
        return
    check(lib().occ_gemm_tn_pair(ctypes.byref(d0), ctypes.byref(d1), stream_ptr()), "occ_gemm_tn_pair")


def colsum(A, a_map, M, N, out, alpha=1.0, a_dtype=0):
    """a_dtype: 0 = f32 (default), 1 = bf16; pass a tensor for A to have it inferred."""
    if hasattr(A, "dtype"):
        a_dtype = 1 if A.dtype == torch.bfloat16 else 0
    check(lib().occ_colsum(_a(A), int(a_dtype), ctypes.byref(a_map), int(M), int(N), _a(out), float(alpha), stream_ptr()), "occ_colsum")


def fill(t, v=0.0):
    check(lib().occ_fill_f32(_a(t), float(v), t.numel(), stream_ptr()), "occ_fill_f32")
    return t


def axpby(a, b, out, alpha=1.0, beta=1.0):
    check(lib().occ_axpby_f32(_a(a), _a(b), _a(out), float(alpha), float(beta), a.numel(), stream_ptr()), "occ_axpby_f32")
    return out


def copy_strided(src, dst, offset, shape4, strides4, accumulate=False):
    sh = np.ascontiguousarray(shape4, dtype=np.int64)
    st = np.ascontiguousarray(strides4, dtype=np.int64)
    check(lib().occ_copy_strided(_a(src), _a(dst), int(offset), ctypes.c_void_p(sh.ctypes.data), ctypes.c_void_p(st.ctypes.data),
                                 int(accumulate), stream_ptr()), "occ_copy_strided")
    return dst


def copy_rows(x, x_map, y, y_map, rows, C, accumulate=False):
    check(lib().occ_copy_rows(_a(x), ctypes.byref(x_map), _a(y), ctypes.byref(y_map), int(rows), int(C), int(accumulate), stream_ptr()),
          "occ_copy_rows")


def act_bwd(dy, y, dx, act):
    check(lib().occ_act_bwd(_a(dy), _a(y), _a(dx), int(act), dy.numel(), stream_ptr()), "occ_act_bwd")
    return dx


def dropout(x, y, mask, p, seed, stream_id, generate):
    check(lib().occ_dropout(_a(x), _a(y), _a(mask), x.numel(), float(p), int(seed), int(stream_id), int(generate), stream_ptr()), "occ_dropout")
    return y


def dropout_step(x, y, mask, p, seed, step_dev, site):
    """Keep-mask drawn on the device from (seed, *step_dev, site): the graph-capturable form of ``dropout(..., generate=True)``."""
    check(lib().occ_dropout_step(_a(x), _a(y), _a(mask), x.numel(), float(p), int(seed), _a(step_dev), int(site), stream_ptr()), "occ_dropout_step")
    return y


def add_u64(counter, v=1):
    check(lib().occ_add_u64(_a(counter), int(v), stream_ptr()), "occ_add_u64")


def stem_pool_fwd(y, out, idx, B, T, F, out_c=1):
    check(lib().occ_stem_pool_fwd(_a(y), _a(out), _a(idx), B, T, F, out_c, stream_ptr()), "occ_stem_pool_fwd")


def stem_pool_bwd(dout, idx, dy, B, T, F, dout_c=1):
    check(lib().occ_stem_pool_bwd(_a(dout), _a(idx), _a(dy), B, T, F, dout_c, stream_ptr()), "occ_stem_pool_bwd")


def bn_stats(x, x_map, rows, C, ws, mean, rstd, run_mean, run_var, nbt, train, momentum=0.1, eps=1e-5):
    check(lib().occ_bn_stats(_a(x), ctypes.byref(x_map), int(rows), int(C), _a(ws), _a(mean), _a(rstd), _a(run_mean), _a(run_var), _a(nbt),
                             float(momentum), float(eps), int(train), stream_ptr()), "occ_bn_stats")


def bn_act_fwd(x, x_map, mean, rstd, gamma, beta, act, y, y_map, rows, C):
    check(lib().occ_bn_act_fwd(_a(x), ctypes.byref(x_map), _a(mean), _a(rstd), _a(gamma), _a(beta), int(act), _a(y), ctypes.byref(y_map),
                               int(rows), int(C), stream_ptr()), "occ_bn_act_fwd")


def bn_act_bwd(dy, dy_map, x, x_map, mean, rstd, gamma, beta, act, dx, dx_map, dgamma, dbeta, ws, sums, rows, C):
    check(lib().occ_bn_act_bwd(_a(dy), ctypes.byref(dy_map), _a(x), ctypes.byref(x_map), _a(mean), _a(rstd), _a(gamma), _a(beta), int(act),
                               _a(dx), ctypes.byref(dx_map), _a(dgamma), _a(dbeta), _a(ws), _a(sums), int(rows), int(C), stream_ptr()),
          "occ_bn_act_bwd")


def softmax_wsum_fwd(x, w, n_outer, inner_n, outer_stride, inner_stride, R, r_stride, C, pos, pos_period, out):
    check(lib().occ_softmax_wsum_fwd(_a(x), _a(w), n_outer, inner_n, outer_stride, inner_stride, R, r_stride, C, _a(pos), pos_period, _a(out),
                                     stream_ptr()), "occ_softmax_wsum_fwd")


def softmax_wsum_bwd(x, w, n_outer, inner_n, outer_stride, inner_stride, R, r_stride, C, dm, dx, dw, accumulate):
    check(lib().occ_softmax_wsum_bwd(_a(x), _a(w), n_outer, inner_n, outer_stride, inner_stride, R, r_stride, C, _a(dm), _a(dx), _a(dw),
                                     int(accumulate), stream_ptr()), "occ_softmax_wsum_bwd")


def pair_mul(x, P, B, N, D):
    check(lib().occ_pair_mul(_a(x), _a(P), B, N, D, stream_ptr()), "occ_pair_mul")


def pair_mul_bwd(dP, x, dx, B, N, D, accumulate):
    check(lib().occ_pair_mul_bwd(_a(dP), _a(x), _a(dx), B, N, D, int(accumulate), stream_ptr()), "occ_pair_mul_bwd")


def gat_softmax(A, aw, B, N, Do, n1, inv_temp, alpha):
    check(lib().occ_gat_softmax(_a(A), _a(aw), B, N, Do, n1, float(inv_temp), _a(alpha), stream_ptr()), "occ_gat_softmax")


def bmm_alpha(alpha, x, out, B, N, D, trans, accumulate):
    check(lib().occ_bmm_alpha(_a(alpha), _a(x), _a(out), B, N, D, int(trans), int(accumulate), stream_ptr()), "occ_bmm_alpha")


def gat_dscore(alpha, dh, x, ds, B, N, D, inv_temp):
    check(lib().occ_gat_dscore(_a(alpha), _a(dh), _a(x), _a(ds), B, N, D, float(inv_temp), stream_ptr()), "occ_gat_dscore")


def gat_core_fwd(x, att_w, att_b, aw3, alpha, h, B, N, D, Do, n1, inv_temp):
    check(lib().occ_gat_core_fwd(_a(x), _a(att_w), _a(att_b), _a(aw3), _a(alpha), _a(h), B, N, D, Do, n1, float(inv_temp), stream_ptr()), "occ_gat_core_fwd")


_GAT_WS = {}


def gat_core_bwd(x, att_w, att_b, aw3, ds, dx, d_att_w, d_att_b, d_aw3, B, N, D, Do, n1):
    need = B * ((N + 15) // 16) * (Do * D + 4 * Do)
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    ws = _GAT_WS.get(key)
    if ws is None or ws.numel() < need:                        # per-workgroup weight-gradient records (summed in order by a second launch)
        ws = _GAT_WS[key] = torch.empty(need, device="cuda", dtype=torch.float32)
    check(lib().occ_gat_core_bwd(_a(x), _a(att_w), _a(att_b), _a(aw3), _a(ds), _a(dx), _a(d_att_w), _a(d_att_b), _a(d_aw3), B, N, D, Do, n1, _a(ws), ws.numel(),
                                 stream_ptr()), "occ_gat_core_bwd")


def gat_dz(A, ds, aw, B, N, Do, n1, daw):
    check(lib().occ_gat_dz(_a(A), _a(ds), _a(aw), B, N, Do, n1, _a(daw), stream_ptr()), "occ_gat_dz")


def master_desc(B, N, D, Do, x, master, master_bstride, w, inv_temp, out, am, agg):
    """w: dict with att_projM.{weight,bias}, att_weightM, proj_with_attM.*, proj_without_attM.* tensors."""
    d = MasterDesc()
    d.B, d.N, d.D, d.Do = B, N, D, Do
    d.x, d.master, d.master_bstride = _ai(x), _ai(master), int(master_bstride)
    d.att_projM_w, d.att_projM_b, d.att_weightM = _ai(w["att_projM.weight"]), _ai(w["att_projM.bias"]), _ai(w["att_weightM"])
    d.proj_with_attM_w, d.proj_with_attM_b = _ai(w["proj_with_attM.weight"]), _ai(w["proj_with_attM.bias"])
    d.proj_without_attM_w, d.proj_without_attM_b = _ai(w["proj_without_attM.weight"]), _ai(w["proj_without_attM.bias"])
    d.inv_temp = float(inv_temp)
    d.out, d.am, d.agg = _ai(out), _ai(am), _ai(agg)
    return d


def master_fwd(d):
    check(lib().occ_master_fwd(ctypes.byref(d), stream_ptr()), "occ_master_fwd")


def master_bwd(d, dout, dx, dx_accumulate, dmaster, dmaster_bstride, g):
    q = MasterGrads()
    q.dout, q.dx, q.dx_accumulate, q.dmaster, q.dmaster_bstride = _ai(dout), _ai(dx), int(dx_accumulate), _ai(dmaster), int(dmaster_bstride)
    q.d_att_projM_w, q.d_att_projM_b, q.d_att_weightM = _ai(g["att_projM.weight"]), _ai(g["att_projM.bias"]), _ai(g["att_weightM"])
    q.d_proj_with_attM_w, q.d_proj_with_attM_b = _ai(g["proj_with_attM.weight"]), _ai(g["proj_with_attM.bias"])
    q.d_proj_without_attM_w, q.d_proj_without_attM_b = _ai(g["proj_without_attM.weight"]), _ai(g["proj_without_attM.bias"])
    check(lib().occ_master_bwd(ctypes.byref(d), ctypes.byref(q), stream_ptr()), "occ_master_bwd")


def graph_pool_fwd(h, mask, p, w, bias, B, N, D, k, out, idx, scores):
    check(lib().occ_graph_pool_fwd(_a(h), _a(mask), float(p), _a(w), _a(bias), B, N, D, k, _a(out), _a(idx), _a(scores), stream_ptr()),
          "occ_graph_pool_fwd")


def graph_pool_bwd(h, mask, p, w, scores, idx, dout, B, N, D, k, dh, dw, dbias):
    check(lib().occ_graph_pool_bwd(_a(h), _a(mask), float(p), _a(w), _a(scores), _a(idx), _a(dout), B, N, D, k, _a(dh), _a(dw), _a(dbias),
                                   stream_ptr()), "occ_graph_pool_bwd")


def readout_desc(B, Nt, Ns, Dg, ncls, T1, T2, S1, S2, M1, M2, masks, p_way, p_last, W, b, emb, logits):
    d = ReadoutDesc()
    d.B, d.Nt, d.Ns, d.Dg, d.n_classes = B, Nt, Ns, Dg, ncls
    d.T1, d.T2, d.S1, d.S2, d.M1, d.M2 = [_ai(t) for t in (T1, T2, S1, S2, M1, M2)]
    m = masks or {}
    d.mask_T1, d.mask_T2, d.mask_S1, d.mask_S2 = _ai(m.get("way_T1")), _ai(m.get("way_T2")), _ai(m.get("way_S1")), _ai(m.get("way_S2"))
    d.mask_M1, d.mask_M2, d.mask_last = _ai(m.get("way_M1")), _ai(m.get("way_M2")), _ai(m.get("last"))
    d.p_way, d.p_last = float(p_way), float(p_last)
    d.out_w, d.out_b, d.emb, d.logits = _ai(W), _ai(b), _ai(emb), _ai(logits)
    return d


def readout_fwd(d):
    check(lib().occ_readout_fwd(ctypes.byref(d), stream_ptr()), "occ_readout_fwd")


def readout_bwd(d, demb, dlogits, dT1, dT2, dS1, dS2, dM1, dM2, dW, db):
    g = ReadoutGrads()
    g.demb, g.dlogits = _ai(demb), _ai(dlogits)
    g.dT1, g.dT2, g.dS1, g.dS2, g.dM1, g.dM2 = [_ai(t) for t in (dT1, dT2, dS1, dS2, dM1, dM2)]
    g.d_out_w, g.d_out_b = _ai(dW), _ai(db)
    check(lib().occ_readout_bwd(ctypes.byref(d), ctypes.byref(g), stream_ptr()), "occ_readout_bwd")
