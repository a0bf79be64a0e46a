"""AASIST back-end + XLS-R front-end on MI355X -- drop-in for ``models/sslassist.py`` (AModel :432-597).

``AModel(args, device)`` keeps the reference's constructor, ``forward(x) -> (emb[B,160], out[B,2])`` and
the state_dict key names (``LL.*``, ``first_bn.*``, ``encoder.{0..5}.0.*``, ``attention.{0,2,3}.*``,
``GAT_layer_{S,T}.*``, ``HtrgGAT_layer_ST{11,12,21,22}.*``, ``pool_*.proj.*``, ``pos_S``, ``master1/2``,
``out_layer.*`` and ``ssl_model.model.*``).  All arithmetic runs in HIP kernels behind libocc_hip.so;
forward and backward are written out explicitly (no autograd tape), every tensor is f32 channels-last
and every convolution is an implicit GEMM over zero-bordered buffers that persist per (B, T).

Reference quirks reproduced: bn1+selu of a residual block is dead code whose
running statistics still move (:409-415); the returned ``emb`` is the dropped-out tensor in train mode
(:591-594); ``master1/2`` enter the first heterogeneous layers un-expanded (:550, 563).
"""
import math

import functools

import torch

from .. import backend_ops as K
from .. import ops
from .._lib import ACT_NONE, ACT_SELU, ACT_TANH, OCC_F32, OCC_F32_AS_BF16, OccError, require_gpu
from ..autograd_bridge import AliasGuard, attach_parameters, run_engine
from ..ops import rowmap
from . import xlsr as xlsr_mod

FILTS = [(1, 32), (32, 32), (32, 64), (64, 64), (64, 64), (64, 64)]      # sslassist.py:438, 457-463
GAT_DIMS = (64, 32)
TEMPS = {"GAT_layer_S": 2.0, "GAT_layer_T": 2.0, "HtrgGAT_layer_ST11": 100.0, "HtrgGAT_layer_ST12": 100.0,
         "HtrgGAT_layer_ST21": 100.0, "HtrgGAT_layer_ST22": 100.0}
P_GAT, P_POOL, P_WAY, P_LAST = 0.2, 0.3, 0.2, 0.5
BN_NAMES = (["first_bn", "first_bn1", "attention.2"] + ["encoder.%d.0.bn2" % i for i in range(6)] +
            ["encoder.%d.0.bn1" % i for i in range(1, 6)] + ["GAT_layer_S.bn", "GAT_layer_T.bn"] +
            ["HtrgGAT_layer_ST%s.bn" % t for t in ("11", "12", "21", "22")])
_BN_C = {"first_bn": 1, "first_bn1": 64, "attention.2": 128, "GAT_layer_S.bn": 64, "GAT_layer_T.bn": 64}
for _i, (_ci, _co) in enumerate(FILTS):
    _BN_C["encoder.%d.0.bn2" % _i] = _co
    if _i > 0:
        _BN_C["encoder.%d.0.bn1" % _i] = _ci
for _t in ("11", "12", "21", "22"):
    _BN_C["HtrgGAT_layer_ST%s.bn" % _t] = 32


def _cpad(c):
    return 4 if c < 4 else c


def backend_param_table():
    """[(reference name, reference shape)] of the trainable tensors (sslassist.py:433-504)."""
    t = []

    def lin(pre, i, o):
        t.append((pre + ".weight", (o, i))); t.append((pre + ".bias", (o,)))

    def bn(pre, c):
        t.append((pre + ".weight", (c,))); t.append((pre + ".bias", (c,)))

    lin("LL", 1024, 128)
    bn("first_bn", 1); bn("first_bn1", 64)
    for i, (ci, co) in enumerate(FILTS):
        pre = "encoder.%d.0" % i
        if i > 0:
            bn(pre + ".bn1", ci)
        t.append((pre + ".conv1.weight", (co, ci, 2, 3))); t.append((pre + ".conv1.bias", (co,)))
        bn(pre + ".bn2", co)
        t.append((pre + ".conv2.weight", (co, co, 2, 3))); t.append((pre + ".conv2.bias", (co,)))
        if ci != co:
            t.append((pre + ".conv_downsample.weight", (co, ci, 1, 3))); t.append((pre + ".conv_downsample.bias", (co,)))
    t.append(("attention.0.weight", (128, 64, 1, 1))); t.append(("attention.0.bias", (128,)))
    bn("attention.2", 128)
    t.append(("attention.3.weight", (64, 128, 1, 1))); t.append(("attention.3.bias", (64,)))
    t += [("pos_S", (1, 42, 64)), ("master1", (1, 1, 64)), ("master2", (1, 1, 64))]
    for pre in ("GAT_layer_S", "GAT_layer_T"):
        lin(pre + ".att_proj", 64, 64); t.append((pre + ".att_weight", (64, 1)))
        lin(pre + ".proj_with_att", 64, 64); lin(pre + ".proj_without_att", 64, 64); bn(pre + ".bn", 64)
    for tag, di in (("ST11", 64), ("ST12", 32), ("ST21", 64), ("ST22", 32)):
        pre = "HtrgGAT_layer_" + tag
        lin(pre + ".proj_type1", di, di); lin(pre + ".proj_type2", di, di)
        lin(pre + ".att_proj", di, 32); lin(pre + ".att_projM", di, 32)
        for w in ("att_weight11", "att_weight22", "att_weight12", "att_weightM"):
            t.append((pre + "." + w, (32, 1)))
        lin(pre + ".proj_with_att", di, 32); lin(pre + ".proj_without_att", di, 32)
        lin(pre + ".proj_with_attM", di, 32); lin(pre + ".proj_without_attM", di, 32); bn(pre + ".bn", 32)
    lin("pool_S.proj", 64, 1); lin("pool_T.proj", 64, 1)
    for n in ("pool_hS1", "pool_hT1", "pool_hS2", "pool_hT2"):
        lin(n + ".proj", 32, 1)
    lin("out_layer", 160, 2)
    return t


_AW_ROW = {"att_weight": 0, "att_weight11": 0, "att_weight22": 1, "att_weight12": 2}


class AasistBackend:
    """Explicit forward/backward engine of the AASIST graph back-end (everything after the SSL features)."""
    graph_safe = True           # a train step makes no host decision and draws its masks from device-side counters: OcTrainer may replay it from a HIP graph

    def __init__(self, params=None, device="cuda", seed=0, compute="f32"):
        """compute: "f32" = exact-f32 MFMA everywhere (parity path); "bf16" = Linear / Conv2d forward and input-gradient
        GEMMs round their f32 operands to bf16 on the way into LDS (bf16 MFMA, f32 accumulate); weight gradients,
        BatchNorm, softmax and every reduction stay f32."""
        require_gpu()
        if compute not in ("f32", "bf16"):
            raise OccError("compute must be 'f32' or 'bf16'")
        self.compute = compute
        import os
        self.fuse_gat = os.environ.get("OCC_GAT_FUSED", "1") != "0"
        # weight gradients: exact-f32 MFMA on the parity path, bf16 MFMA (f32 accumulate) in bf16 compute mode
        self._tn = functools.partial(K.gemm_tn, bf16_mfma=(compute == "bf16"))
        self.device = torch.device(device)
        self.table = backend_param_table()
        # ---- flat parameter / gradient storage (one Adam tensor, one all-reduce bucket) ----
        self.slots = {}          # name -> (offset, internal_shape)
        off = 0
        for name, shp in self.table:
            leaf = name.rsplit(".", 1)[-1]
            if leaf in _AW_ROW:
                layer = name.rsplit(".", 1)[0]
                key = layer + ".aw3"
                if key not in self.slots:
                    self.slots[key] = (off, (3, shp[0]))
                    off += 3 * shp[0]
                continue
            if len(shp) == 4 and shp[2:] != (1, 1):
                ishp = (shp[0], shp[2], shp[3], _cpad(shp[1]))
            elif len(shp) == 4:
                ishp = (shp[0], shp[1])
            else:
                ishp = tuple(shp)
            n = 1
            for d in ishp:
                n *= d
            self.slots[name] = (off, ishp)
            off += (n + 3) // 4 * 4                       # keep every tensor 16-byte aligned
        self.n_flat = off
        self.P = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.G = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.p = {k: self._view(self.P, k) for k in self.slots}
        self.g = {k: self._view(self.G, k) for k in self.slots}
        self.buf = {}
        for bn in BN_NAMES:
            c = _BN_C[bn]
            self.buf[bn + ".running_mean"] = torch.zeros(c, device=self.device)
            self.buf[bn + ".running_var"] = torch.ones(c, device=self.device)
            self.buf[bn + ".num_batches_tracked"] = torch.zeros(1, device=self.device, dtype=torch.int64)
        self.bn_ws = torch.empty(512 * 256 * 2, device=self.device, dtype=torch.float64)
        self.bn_sums = torch.empty(512, device=self.device, dtype=torch.float32)
        self._ws = {}
        self.ctx = None
        self.rng_seed = seed
        self.rng_step = 0                       # host mirror of the device counter below (one per train-mode forward)
        self.rng_step_dev = torch.zeros(1, device=self.device, dtype=torch.int64)
        self.load_reference_params(params if params is not None else synthetic_backend_params(seed), strict_buffers=False)

    def _view(self, flat, key):
        off, shp = self.slots[key]
        n = 1
        for d in shp:
            n *= d
        return flat[off:off + n].view(shp)

    # ---------------------------------------------------------------- checkpoint layout <-> internal --
    def load_reference_params(self, sd, strict=True, strict_buffers=None):
        strict_buffers = strict if strict_buffers is None else strict_buffers      # running statistics: a fresh model starts from the defaults
        sd = {k: v for k, v in sd.items() if not k.startswith("ssl_model.")}
        with torch.no_grad():
            for name, shp in self.table:
                if name not in sd:
                    if not strict:
                        continue
                    raise OccError("AASIST state_dict lacks %s" % name)
                v = sd[name].detach().to(self.device, torch.float32)
                if tuple(v.shape) != tuple(shp):
                    raise OccError("shape of %s is %s, expected %s" % (name, tuple(v.shape), tuple(shp)))
                leaf = name.rsplit(".", 1)[-1]
                if leaf in _AW_ROW:
                    self.p[name.rsplit(".", 1)[0] + ".aw3"][_AW_ROW[leaf]].copy_(v.reshape(-1))
                elif len(shp) == 4 and shp[2:] != (1, 1):
                    dst = self.p[name]
                    dst.zero_()
                    dst[..., : shp[1]].copy_(v.permute(0, 2, 3, 1))
                else:
                    self.p[name].copy_(v.reshape(self.p[name].shape))
            for k in self.buf:
                if k in sd:
                    self.buf[k].copy_(sd[k].to(self.device).reshape(self.buf[k].shape))
                elif strict_buffers:
                    raise OccError("AASIST state_dict lacks buffer %s" % k)

    def state_dict(self):
        out = {}
        for name, shp in self.table:
            leaf = name.rsplit(".", 1)[-1]
            if leaf in _AW_ROW:
                out[name] = self.p[name.rsplit(".", 1)[0] + ".aw3"][_AW_ROW[leaf]].detach().clone().reshape(shp)
            elif len(shp) == 4 and shp[2:] != (1, 1):
                out[name] = self.p[name][..., : shp[1]].permute(0, 3, 1, 2).contiguous().clone()
            else:
                out[name] = self.p[name].detach().clone().reshape(shp)
        for k, v in self.buf.items():
            out[k] = v.clone().reshape(()) if k.endswith("num_batches_tracked") else v.clone()
        return out

    def ref_views(self, flat):
        """{reference name: VIEW of ``flat`` (self.P, self.G or a buffer of the same size) in the reference's shape}: conv weights are
        channels-last (and channel-padded) inside, the three typed attention weights of a layer are rows of one [3, Do] slot."""
        out = {}
        for name, shp in self.table:
            leaf = name.rsplit(".", 1)[-1]
            if leaf in _AW_ROW:
                out[name] = self._view(flat, name.rsplit(".", 1)[0] + ".aw3")[_AW_ROW[leaf]].view(shp)
            elif len(shp) == 4 and shp[2:] != (1, 1):
                out[name] = self._view(flat, name)[..., : shp[1]].permute(0, 3, 1, 2)
            else:
                out[name] = self._view(flat, name).view(shp)
        return out

    def grad_dict(self):
        """Gradients in the reference's parameter layout (for tests / interchange)."""
        out = {}
        for name, shp in self.table:
            leaf = name.rsplit(".", 1)[-1]
            if leaf in _AW_ROW:
                out[name] = self.g[name.rsplit(".", 1)[0] + ".aw3"][_AW_ROW[leaf]].detach().clone().reshape(shp)
            elif len(shp) == 4 and shp[2:] != (1, 1):
                out[name] = self.g[name][..., : shp[1]].permute(0, 3, 1, 2).contiguous().clone()
            else:
                out[name] = self.g[name].detach().clone().reshape(shp)
        return out

    # ------------------------------------------------------------------------------ workspaces --
    def _e(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, device=self.device, dtype=dtype)

    def _workspace(self, B, T):
        key = (B, T)
        if key not in self._ws:
            W = T // 3
            if W < 1:
                raise OccError("T=%d frames is too short for the AASIST stem" % T)
            Wp = W + 2
            ws = {"W": W, "Wp": Wp}
            z = lambda *s: torch.zeros(*s, device=self.device, dtype=torch.float32)
            for i, (ci, co) in enumerate(FILTS):
                ws["X%d" % i] = z(B, 44, Wp, _cpad(ci))       # block input, 1-pixel zero border
                ws["Y%d" % i] = z(B, 43, Wp, co)              # selu(bn2(conv1)) with zero columns
                ws["D1_%d" % i] = z(B, 43, Wp, co)            # grad wrt conv1 output
                ws["D%d" % i] = z(B, 44, Wp, co)              # grad wrt block output
            self._ws[key] = ws
            while len(self._ws) > 12:                  # variable-length scoring: keep the most recent shapes only
                self._ws.pop(next(iter(self._ws)))
        else:
            self._ws[key] = self._ws.pop(key)
        return self._ws[key]

    # ---------------------------------------------------------------------------------- helpers --
    def _gd(self, Kd, *strides):
        """GEMM operand mode: bf16 compute needs K and every operand stride in multiples of 8 elements."""
        if self.compute == "bf16" and Kd % 8 == 0 and all(int(v) % 8 == 0 for v in strides):
            return OCC_F32_AS_BF16
        return OCC_F32

    def _bn(self, name, x, x_map, rows, C, train, want_out=True):
        mean, rstd = (self._e(C), self._e(C)) if want_out else (None, None)
        K.bn_stats(x, x_map, rows, C, self.bn_ws, mean, rstd, self.buf[name + ".running_mean"], self.buf[name + ".running_var"],
                   self.buf[name + ".num_batches_tracked"], train)
        return mean, rstd

    def _lin(self, x, M, Kd, name, N, act=ACT_NONE, R=None, out=None, a_map=None, c_addr=None, c_map=None):
        out = out if out is not None else self._e(M, N)
        am = a_map or rowmap(M, 0, Kd)
        ops.gemm_raw(M, N, Kd, x, am, self.p[name + ".weight"], Kd, c_addr if c_addr is not None else out,
                     c_map or rowmap(M, 0, N), OCC_F32, self._gd(Kd, am.row_stride, am.batch_stride), bias=self.p[name + ".bias"], act=act,
                     R=R, r_map=None if R is None else rowmap(M, 0, N), r_dtype=OCC_F32)
        return out

    def _lin_bwd(self, dy, dy_map, x, x_map, M, Kd, name, N, dx=None, dx_map=None, dx_R=None, dx_rmap=None, need_dx=True):
        """dW += dy^T x, db += colsum(dy), dx = dy W (+R).  dy rows [N] via dy_map, x rows [Kd] via x_map."""
        self._tn(M, N, Kd, dy, dy_map, x, x_map, self.g[name + ".weight"], Kd, colsum_out=self.g[name + ".bias"])
        if not need_dx:
            return None
        wt = self._e(Kd, N)
        K.copy_strided(self.p[name + ".weight"], wt, 0, (1, 1, Kd, N), (0, 0, 1, Kd))
        out = dx if dx is not None else self._e(M, Kd)
        ops.gemm_raw(M, Kd, N, dy, dy_map, wt, N, out, dx_map or rowmap(M, 0, Kd), OCC_F32, self._gd(N, dy_map.row_stride, dy_map.batch_stride),
                     R=dx_R, r_map=dx_rmap, r_dtype=OCC_F32)
        return out

    def _drop(self, x, site, p, train, masks):
        """Returns (y, mask or None).  masks: None -> draw with Philox (train only); dict -> injected keep-masks."""
        if not train:
            return x, None
        if masks is not None:
            if site not in masks:
                return x, None
            m = masks[site].to(self.device, torch.uint8).contiguous()
            y = self._e(*x.shape)
            K.dropout(x, y, m, p, 0, 0, generate=False)
            return y, m
        m = torch.empty(x.shape, device=self.device, dtype=torch.uint8)
        y = self._e(*x.shape)
        self._site_id += 1
        K.dropout_step(x, y, m, p, self.rng_seed, self.rng_step_dev, self._site_id)
        return y, m

    def _drop_mask_only(self, shape, site, p, train, masks):
        """Keep-mask for sites whose consumer applies it itself (graph pool, read-out)."""
        if not train:
            return None
        if masks is not None:
            return masks[site].to(self.device, torch.uint8).contiguous() if site in masks else None
        m = torch.empty(shape, device=self.device, dtype=torch.uint8)
        dummy = self._e(*shape)
        self._site_id += 1
        K.dropout_step(dummy, dummy, m, p, self.rng_seed, self.rng_step_dev, self._site_id)
        return m

    # ------------------------------------------------------------------------- conv geometry --
    @staticmethod
    def _maps(W, Wp):
        m = {}
        m["x_win43"] = lambda c: rowmap(43 * W, 44 * Wp * c, c, W, Wp * c)     # conv1 windows over X (44 rows)
        m["x_in42"] = lambda c: rowmap(42 * W, 44 * Wp * c, c, W, Wp * c)      # interior / row windows of X or D (44 rows)
        m["y_win42"] = lambda c: rowmap(42 * W, 43 * Wp * c, c, W, Wp * c)     # conv2 windows over Y / D1 (43 rows)
        m["y_in43"] = lambda c: rowmap(43 * W, 43 * Wp * c, c, W, Wp * c)      # interior of Y / D1
        m["d_win43"] = lambda c: rowmap(43 * W, 44 * Wp * c, c, W, Wp * c)     # conv2 dgrad windows over D (44 rows)
        return m

    # ================================================================================== forward ==
    def forward(self, feats, train=False, masks=None):
        """feats f32 [B,T,1024] on the GPU -> (emb [B,160], logits [B,2]).  train=True saves the tape."""
        feats = feats.to(self.device, torch.float32).contiguous()
        B, T, Fd = feats.shape
        ws = self._workspace(B, T)
        W, Wp = ws["W"], ws["Wp"]
        mp = self._maps(W, Wp)
        R = B * 42 * W
        c = {"B": B, "T": T, "feats": feats, "train": train}
        self._site_id = 0
        p, es = self.p, 4
        # ---- LL + stem ---------------------------------------------------------------------
        c["ll"] = self._lin(feats.view(B * T, Fd), B * T, Fd, "LL", 128)
        c["pool"], c["pool_idx"] = self._e(R), torch.empty(R, device=self.device, dtype=torch.uint8)
        K.stem_pool_fwd(c["ll"], c["pool"], c["pool_idx"], B, T, 128)
        c["bn.first_bn"] = self._bn("first_bn", c["pool"], rowmap(R, 0, 1), R, 1, train)
        X0 = ws["X0"]
        K.bn_act_fwd(c["pool"], rowmap(R, 0, 1), *c["bn.first_bn"], p["first_bn.weight"], p["first_bn.bias"], ACT_SELU,
                     X0.data_ptr() + (Wp + 1) * 4 * es, mp["x_in42"](4), R, 1)
        # ---- residual blocks ---------------------------------------------------------------
        enc = self._e(R, 64)
        for i, (ci0, co) in enumerate(FILTS):
            ci = _cpad(ci0)
            pre = "encoder.%d.0" % i
            X, Y = ws["X%d" % i], ws["Y%d" % i]
            if i > 0 and train:        # dead bn1: only its running statistics move (sslassist.py:409-415)
                self._bn(pre + ".bn1", X.data_ptr() + (Wp + 1) * ci * es, mp["x_in42"](ci), R, ci, True, want_out=False)
            R1 = B * 43 * W
            o1 = self._e(R1, co)
            ops.gemm_raw(R1, co, 6 * ci, X, mp["x_win43"](ci), p[pre + ".conv1.weight"], 6 * ci, o1, rowmap(R1, 0, co), OCC_F32, self._gd(3 * ci, ci),
                         bias=p[pre + ".conv1.bias"], a_seg=(2, 3 * ci, Wp * ci))
            c["o1_%d" % i] = o1
            c["bn2_%d" % i] = self._bn(pre + ".bn2", o1, rowmap(R1, 0, co), R1, co, train)
            K.bn_act_fwd(o1, rowmap(R1, 0, co), *c["bn2_%d" % i], p[pre + ".bn2.weight"], p[pre + ".bn2.bias"], ACT_SELU,
                         Y.data_ptr() + co * es, mp["y_in43"](co), R1, co)
            if ci0 != co:
                ident = self._e(R, co)
                ops.gemm_raw(R, co, 3 * ci, X.data_ptr() + Wp * ci * es, mp["x_in42"](ci), p[pre + ".conv_downsample.weight"], 3 * ci,
                             ident, rowmap(R, 0, co), OCC_F32, self._gd(3 * ci, ci), bias=p[pre + ".conv_downsample.bias"])
                r_addr, r_map = ident, rowmap(R, 0, co)
            else:
                r_addr, r_map = X.data_ptr() + (Wp + 1) * ci * es, mp["x_in42"](ci)
            if i < 5:
                c_addr, c_map = ws["X%d" % (i + 1)].data_ptr() + (Wp + 1) * co * es, mp["x_in42"](co)
            else:
                c_addr, c_map = enc, rowmap(R, 0, co)
            ops.gemm_raw(R, co, 6 * co, Y, mp["y_win42"](co), p[pre + ".conv2.weight"], 6 * co, c_addr, c_map, OCC_F32, self._gd(3 * co, co),
                         bias=p[pre + ".conv2.bias"], a_seg=(2, 3 * co, Wp * co), R=r_addr, r_map=r_map, r_dtype=OCC_F32)
        c["enc"] = enc
        c["bn.first_bn1"] = self._bn("first_bn1", enc, rowmap(R, 0, 64), R, 64, train)
        xa = self._e(R, 64)
        K.bn_act_fwd(enc, rowmap(R, 0, 64), *c["bn.first_bn1"], p["first_bn1.weight"], p["first_bn1.bias"], ACT_SELU, xa, rowmap(R, 0, 64), R, 64)
        c["xa"] = xa
        # ---- attention map (1x1 convs) ------------------------------------------------------
        a1 = self._lin(xa, R, 64, "attention.0", 128, act=ACT_SELU)
        c["a1"] = a1
        c["bn.attention.2"] = self._bn("attention.2", a1, rowmap(R, 0, 128), R, 128, train)
        a2 = self._e(R, 128)
        K.bn_act_fwd(a1, rowmap(R, 0, 128), *c["bn.attention.2"], p["attention.2.weight"], p["attention.2.bias"], ACT_NONE, a2, rowmap(R, 0, 128), R, 128)
        c["a2"] = a2
        wt = self._lin(a2, R, 128, "attention.3", 64)
        c["wt"] = wt
        eS, eT = self._e(B, 42, 64), self._e(B, W, 64)
        K.softmax_wsum_fwd(xa, wt, B * 42, 1, W * 64, 0, W, 64, 64, p["pos_S"], 42, eS)            # softmax over W per (b,h)
        K.softmax_wsum_fwd(xa, wt, B * W, W, 42 * W * 64, 64, 42, W * 64, 64, None, 1, eT)         # softmax over H per (b,w)
        c["eS"], c["eT"] = eS, eT
        # ---- graph attention + pooling --------------------------------------------------------
        gS = self._gat_fwd("GAT_layer_S", eS, B, 42, c, train, masks)
        oS = self._pool_fwd("pool_S", gS, B, 42, 64, c, train, masks)
        gT = self._gat_fwd("GAT_layer_T", eT, B, W, c, train, masks)
        oT = self._pool_fwd("pool_T", gT, B, W, 64, c, train, masks)
        nT, nS = max(int(W * 0.5), 1), 21
        outs = {}
        for tag in ("1", "2"):
            m0 = p["master%s" % tag]
            T1, S1, M1 = self._htrg_fwd("HtrgGAT_layer_ST%s1" % tag, oT, oS, m0, 0, B, nT, nS, 64, c, train, masks)
            S1p = self._pool_fwd("pool_hS%s" % tag, S1, B, nS, 32, c, train, masks)
            T1p = self._pool_fwd("pool_hT%s" % tag, T1, B, nT, 32, c, train, masks)
            nT2, nS2 = max(int(nT * 0.5), 1), max(int(nS * 0.5), 1)
            Ta, Sa, Ma = self._htrg_fwd("HtrgGAT_layer_ST%s2" % tag, T1p, S1p, M1, 32, B, nT2, nS2, 32, c, train, masks)
            outs["T" + tag] = K.axpby(T1p, Ta, self._e(B, nT2, 32))
            outs["S" + tag] = K.axpby(S1p, Sa, self._e(B, nS2, 32))
            outs["M" + tag] = K.axpby(M1, Ma, self._e(B, 32))
        c["nT"], c["nS"], c["nT2"], c["nS2"] = nT, nS, nT2, nS2
        # ---- read-out ---------------------------------------------------------------------------
        rm = {}
        for site, key in (("way_T1", "T1"), ("way_T2", "T2"), ("way_S1", "S1"), ("way_S2", "S2"), ("way_M1", "M1"), ("way_M2", "M2")):
            m = self._drop_mask_only(outs[key].shape, site, P_WAY, train, masks)
            if m is not None:
                rm[site] = m
        m = self._drop_mask_only((B, 160), "last", P_LAST, train, masks)
        if m is not None:
            rm["last"] = m
        emb, logits = self._e(B, 160), self._e(B, 2)
        c["rd"] = K.readout_desc(B, nT2, nS2, 32, 2, outs["T1"], outs["T2"], outs["S1"], outs["S2"], outs["M1"], outs["M2"], rm, P_WAY, P_LAST,
                                 p["out_layer.weight"], p["out_layer.bias"], emb, logits)
        c["rd_keep"] = (outs, rm, emb, logits)
        K.readout_fwd(c["rd"])
        self.ctx = c if train else None
        if train:
            self.rng_step += 1
            K.add_u64(self.rng_step_dev, 1)     # on the stream, so a captured step advances it on every replay
        return emb, logits

    # ---- GraphAttentionLayer (sslassist.py:58-151) ------------------------------------------------
    def _att_core_fwd(self, pre, xd, B, N, D, Do, n1, c):
        """pairwise product -> att_proj+tanh -> typed score -> softmax -> alpha@x."""
        if self._fused_gat(N, D, Do):         # bf16-compute mode: one kernel, no [B,N,N,D] / [B,N,N,Do] tensors (csrc/gat_fused.hip)
            alpha, h = self._e(B, N, N), self._e(B, N, D)
            K.gat_core_fwd(xd, self.p[pre + ".att_proj.weight"], self.p[pre + ".att_proj.bias"], self.p[pre + ".aw3"], alpha, h, B, N, D, Do, n1, 1.0 / TEMPS[pre])
            c[pre + ".P"], c[pre + ".A"], c[pre + ".alpha"], c[pre + ".h"] = None, None, alpha, h
            return h
        P = self._e(B * N * N, D)
        K.pair_mul(xd, P, B, N, D)
        A = self._lin(P, B * N * N, D, pre + ".att_proj", Do, act=ACT_TANH)
        alpha = self._e(B, N, N)
        K.gat_softmax(A, self.p[pre + ".aw3"], B, N, Do, n1, 1.0 / TEMPS[pre], alpha)
        h = self._e(B, N, D)
        K.bmm_alpha(alpha, xd, h, B, N, D, 0, 0)
        c[pre + ".P"], c[pre + ".A"], c[pre + ".alpha"], c[pre + ".h"] = P, A, alpha, h
        return h

    def _fused_gat(self, N, D, Do):
        """The fused attention-core kernels: bf16-compute mode (the exact-f32 parity mode keeps the unfused f32 path), the AASIST layer
        widths, node counts whose backward LDS image fits (N <= 96: 4 s utterances give 42 / 66 / 54 / 26)."""
        return self.compute == "bf16" and self.fuse_gat and (D, Do) in ((64, 64), (64, 32), (32, 32)) and N <= 96

    def _gat_fwd(self, pre, x, B, N, c, train, masks):
        D = Do = 64
        xd, m = self._drop(x, pre, P_GAT, train, masks)
        c[pre + ".xd"], c[pre + ".mask"] = xd, m
        h = self._att_core_fwd(pre, xd, B, N, D, Do, N, c)
        y = self._lin(h.view(B * N, D), B * N, D, pre + ".proj_with_att", Do)
        y = self._lin(xd.view(B * N, D), B * N, D, pre + ".proj_without_att", Do, R=y, out=y)
        c[pre + ".y"] = y
        c[pre + ".bn"] = self._bn(pre + ".bn", y, rowmap(B * N, 0, Do), B * N, Do, train)
        out = self._e(B, N, Do)
        K.bn_act_fwd(y, rowmap(B * N, 0, Do), *c[pre + ".bn"], self.p[pre + ".bn.weight"], self.p[pre + ".bn.bias"], ACT_SELU, out,
                     rowmap(B * N, 0, Do), B * N, Do)
        return out

    def _att_core_bwd(self, pre, dh, dxd, B, N, D, Do, n1, c):
        """adds the attention-path gradient into dxd; accumulates att_proj / att_weight grads."""
        P, A, alpha, xd = c[pre + ".P"], c[pre + ".A"], c[pre + ".alpha"], c[pre + ".xd"]
        K.bmm_alpha(alpha, dh, dxd, B, N, D, 1, 1)
        ds = self._e(B, N, N)
        K.gat_dscore(alpha, dh, xd, ds, B, N, D, 1.0 / TEMPS[pre])
        if P is None:                          # fused forward: z is recomputed in the fused backward, nothing of size N*N*D was kept
            K.gat_core_bwd(xd, self.p[pre + ".att_proj.weight"], self.p[pre + ".att_proj.bias"], self.p[pre + ".aw3"], ds, dxd, self.g[pre + ".att_proj.weight"],
                           self.g[pre + ".att_proj.bias"], self.g[pre + ".aw3"], B, N, D, Do, n1)
            return
        K.gat_dz(A, ds, self.p[pre + ".aw3"], B, N, Do, n1, self.g[pre + ".aw3"])          # A <- dZ
        M = B * N * N
        dP = self._lin_bwd(A, rowmap(M, 0, Do), P, rowmap(M, 0, D), M, D, pre + ".att_proj", Do)
        K.pair_mul_bwd(dP, xd, dxd, B, N, D, 1)

    def _gat_bwd(self, pre, dout, B, N, c):
        D = Do = 64
        M = B * N
        y, xd = c[pre + ".y"], c[pre + ".xd"]
        dy = self._e(M, Do)
        K.bn_act_bwd(dout, rowmap(M, 0, Do), y, rowmap(M, 0, Do), *c[pre + ".bn"], self.p[pre + ".bn.weight"], self.p[pre + ".bn.bias"], ACT_SELU,
                     dy, rowmap(M, 0, Do), self.g[pre + ".bn.weight"], self.g[pre + ".bn.bias"], self.bn_ws, self.bn_sums, M, Do)
        dxd = self._lin_bwd(dy, rowmap(M, 0, Do), xd, rowmap(M, 0, D), M, D, pre + ".proj_without_att", Do)
        dh = self._lin_bwd(dy, rowmap(M, 0, Do), c[pre + ".h"], rowmap(M, 0, D), M, D, pre + ".proj_with_att", Do)
        self._att_core_bwd(pre, dh, dxd, B, N, D, Do, N, c)
        return self._drop_bwd(dxd, c[pre + ".mask"], P_GAT)

    def _drop_bwd(self, dy, mask, p):
        if mask is None:
            return dy
        dx = self._e(*dy.shape)
        K.dropout(dy, dx, mask, p, 0, 0, generate=False)
        return dx

    # ---- GraphPool (sslassist.py:332-368) ----------------------------------------------------------
    def _pool_fwd(self, pre, h, B, N, D, c, train, masks):
        k = max(int(N * 0.5), 1)
        m = self._drop_mask_only((B, N, D), pre, P_POOL, train, masks)
        out, idx, sc = self._e(B, k, D), torch.empty(B, k, device=self.device, dtype=torch.int32), self._e(B, N)
        K.graph_pool_fwd(h, m, P_POOL, self.p[pre + ".proj.weight"], self.p[pre + ".proj.bias"], B, N, D, k, out, idx, sc)
        c[pre] = (h, m, idx, sc, N, D, k)
        return out

    def _pool_bwd(self, pre, dout, B, c):
        h, m, idx, sc, N, D, k = c[pre]
        dh = self._e(B, N, D)
        K.graph_pool_bwd(h, m, P_POOL, self.p[pre + ".proj.weight"], sc, idx, dout, B, N, D, k, dh, self.g[pre + ".proj.weight"],
                         self.g[pre + ".proj.bias"])
        return dh

    # ---- HtrgGraphAttentionLayer (sslassist.py:154-329) --------------------------------------------
    def _htrg_fwd(self, pre, x1, x2, master, m_bstride, B, N1, N2, Din, c, train, masks):
        Do, N = 32, N1 + N2
        xc = self._e(B, N, Din)
        self._lin(x1.view(B * N1, Din), B * N1, Din, pre + ".proj_type1", Din, c_addr=xc, c_map=rowmap(N1, N * Din, Din))
        self._lin(x2.view(B * N2, Din), B * N2, Din, pre + ".proj_type2", Din, c_addr=xc.data_ptr() + N1 * Din * 4, c_map=rowmap(N2, N * Din, Din))
        xd, m = self._drop(xc, pre, P_GAT, train, masks)
        c[pre + ".x1"], c[pre + ".x2"], c[pre + ".xd"], c[pre + ".mask"] = x1, x2, xd, m
        h = self._att_core_fwd(pre, xd, B, N, Din, Do, N1, c)
        w = {k: self.p[pre + "." + k] for k in ("att_projM.weight", "att_projM.bias", "proj_with_attM.weight", "proj_with_attM.bias",
                                                "proj_without_attM.weight", "proj_without_attM.bias")}
        w["att_weightM"] = self.p[pre + ".att_weightM"]
        mo, am, agg = self._e(B, Do), self._e(B, N), self._e(B, Din)
        md = K.master_desc(B, N, Din, Do, xd, master, m_bstride, w, 1.0 / TEMPS[pre], mo, am, agg)
        K.master_fwd(md)
        c[pre + ".md"], c[pre + ".md_keep"] = md, (master, mo, am, agg)
        y = self._lin(h.view(B * N, Din), B * N, Din, pre + ".proj_with_att", Do)
        y = self._lin(xd.view(B * N, Din), B * N, Din, pre + ".proj_without_att", Do, R=y, out=y)
        c[pre + ".y"] = y
        c[pre + ".bn"] = self._bn(pre + ".bn", y, rowmap(B * N, 0, Do), B * N, Do, train)
        o1, o2 = self._e(B, N1, Do), self._e(B, N2, Do)
        g, b_ = self.p[pre + ".bn.weight"], self.p[pre + ".bn.bias"]
        K.bn_act_fwd(y, rowmap(N1, N * Do, Do), *c[pre + ".bn"], g, b_, ACT_SELU, o1, rowmap(B * N1, 0, Do), B * N1, Do)
        K.bn_act_fwd(y.data_ptr() + N1 * Do * 4, rowmap(N2, N * Do, Do), *c[pre + ".bn"], g, b_, ACT_SELU, o2, rowmap(B * N2, 0, Do), B * N2, Do)
        c[pre + ".dims"] = (N1, N2, Din)
        return o1, o2, mo

    def _htrg_bwd(self, pre, d1, d2, dmo, dmaster, dm_bstride, B, c):
        """d1 [B,N1,32], d2 [B,N2,32], dmo [B,32] -> (dx1, dx2); the master-input gradient is added into dmaster."""
        N1, N2, Din = c[pre + ".dims"]
        Do, N = 32, N1 + N2
        M = B * N
        dyc = self._e(B, N, Do)
        K.copy_rows(d1, rowmap(B * N1, 0, Do), dyc, rowmap(N1, N * Do, Do), B * N1, Do)
        K.copy_rows(d2, rowmap(B * N2, 0, Do), dyc.data_ptr() + N1 * Do * 4, rowmap(N2, N * Do, Do), B * N2, Do)
        y, xd = c[pre + ".y"], c[pre + ".xd"]
        dy = self._e(M, Do)
        K.bn_act_bwd(dyc, rowmap(M, 0, Do), y, rowmap(M, 0, Do), *c[pre + ".bn"], self.p[pre + ".bn.weight"], self.p[pre + ".bn.bias"], ACT_SELU,
                     dy, rowmap(M, 0, Do), self.g[pre + ".bn.weight"], self.g[pre + ".bn.bias"], self.bn_ws, self.bn_sums, M, Do)
        dxd = self._lin_bwd(dy, rowmap(M, 0, Do), xd, rowmap(M, 0, Din), M, Din, pre + ".proj_without_att", Do)
        dh = self._lin_bwd(dy, rowmap(M, 0, Do), c[pre + ".h"], rowmap(M, 0, Din), M, Din, pre + ".proj_with_att", Do)
        g = {k: self.g[pre + "." + k] for k in ("att_projM.weight", "att_projM.bias", "proj_with_attM.weight", "proj_with_attM.bias",
                                                "proj_without_attM.weight", "proj_without_attM.bias")}
        g["att_weightM"] = self.g[pre + ".att_weightM"]
        K.master_bwd(c[pre + ".md"], dmo, dxd, 1, dmaster, dm_bstride, g)
        self._att_core_bwd(pre, dh, dxd, B, N, Din, Do, N1, c)
        dxc = self._drop_bwd(dxd, c[pre + ".mask"], P_GAT)
        x1, x2 = c[pre + ".x1"], c[pre + ".x2"]
        dx1 = self._lin_bwd(dxc, rowmap(N1, N * Din, Din), x1, rowmap(B * N1, 0, Din), B * N1, Din, pre + ".proj_type1", Din)
        dx2 = self._lin_bwd(dxc.data_ptr() + N1 * Din * 4, rowmap(N2, N * Din, Din), x2, rowmap(B * N2, 0, Din), B * N2, Din, pre + ".proj_type2", Din)
        return dx1.view(B, N1, Din), dx2.view(B, N2, Din)

    # ================================================================================= backward ==
    def zero_grad(self):
        K.fill(self.G, 0.0)

    def backward(self, demb, dlogits, want_dfeats=False):
        """Accumulates parameter gradients into ``self.G`` (reference layouts via grad_dict()).
        demb may be None (loss without a compactness term)."""
        c = self.ctx
        if c is None:
            raise OccError("backward() needs a preceding forward(train=True)")
        B, T = c["B"], c["T"]
        ws = self._workspace(B, T)
        W, Wp = ws["W"], ws["Wp"]
        mp = self._maps(W, Wp)
        R, es, p, g = B * 42 * W, 4, self.p, self.g
        nT, nS, nT2, nS2 = c["nT"], c["nS"], c["nT2"], c["nS2"]
        outs = c["rd_keep"][0]
        dT = {k: self._e(*outs[k].shape) for k in ("T1", "T2", "S1", "S2", "M1", "M2")}
        K.readout_bwd(c["rd"], demb, dlogits, dT["T1"], dT["T2"], dT["S1"], dT["S2"], dT["M1"], dT["M2"], g["out_layer.weight"], g["out_layer.bias"])
        d_oT, d_oS = self._e(B, nT, 64), self._e(B, nS, 64)
        first = True
        for tag in ("1", "2"):
            dTf, dSf, dMf = dT["T" + tag], dT["S" + tag], dT["M" + tag]
            dM1 = self._e(B, 32)
            K.axpby(dMf, None, dM1, 1.0, 0.0)                                        # dM1 = dMf (+ ST?2 master-input grad below)
            dT1p_a, dS1p_a = self._htrg_bwd("HtrgGAT_layer_ST%s2" % tag, dTf, dSf, dMf, dM1, 32, B, c)
            dT1p = K.axpby(dTf, dT1p_a, self._e(B, nT2, 32))
            dS1p = K.axpby(dSf, dS1p_a, self._e(B, nS2, 32))
            dT1 = self._pool_bwd("pool_hT%s" % tag, dT1p, B, c)
            dS1 = self._pool_bwd("pool_hS%s" % tag, dS1p, B, c)
            doT_k, doS_k = self._htrg_bwd("HtrgGAT_layer_ST%s1" % tag, dT1, dS1, dM1, g["master%s" % tag], 0, B, c)
            if first:
                K.axpby(doT_k, None, d_oT, 1.0, 0.0); K.axpby(doS_k, None, d_oS, 1.0, 0.0)
                first = False
            else:
                K.axpby(d_oT, doT_k, d_oT); K.axpby(d_oS, doS_k, d_oS)
        d_gT = self._pool_bwd("pool_T", d_oT, B, c)
        d_eT = self._gat_bwd("GAT_layer_T", d_gT, B, W, c)
        d_gS = self._pool_bwd("pool_S", d_oS, B, c)
        d_eS = self._gat_bwd("GAT_layer_S", d_gS, B, 42, c)
        K.colsum(d_eS, rowmap(B, 0, 42 * 64), B, 42 * 64, g["pos_S"])
        xa, wt = c["xa"], c["wt"]
        dxa, dwt = self._e(R, 64), self._e(R, 64)
        K.softmax_wsum_bwd(xa, wt, B * 42, 1, W * 64, 0, W, 64, 64, d_eS, dxa, dwt, 0)
        K.softmax_wsum_bwd(xa, wt, B * W, W, 42 * W * 64, 64, 42, W * 64, 64, d_eT, dxa, dwt, 1)
        # ---- attention 1x1 convs ---------------------------------------------------------------
        da2 = self._lin_bwd(dwt, rowmap(R, 0, 64), c["a2"], rowmap(R, 0, 128), R, 128, "attention.3", 64)
        da1 = self._e(R, 128)
        K.bn_act_bwd(da2, rowmap(R, 0, 128), c["a1"], rowmap(R, 0, 128), *c["bn.attention.2"], p["attention.2.weight"], p["attention.2.bias"], ACT_NONE,
                     da1, rowmap(R, 0, 128), g["attention.2.weight"], g["attention.2.bias"], self.bn_ws, self.bn_sums, R, 128)
        K.act_bwd(da1, c["a1"], da1, ACT_SELU)
        self._lin_bwd(da1, rowmap(R, 0, 128), xa, rowmap(R, 0, 64), R, 64, "attention.0", 128, dx=dxa, dx_R=dxa, dx_rmap=rowmap(R, 0, 64))
        # ---- first_bn1 -> grad wrt encoder output, written into D5's interior ---------------------
        D = ws["D5"]
        K.bn_act_bwd(dxa, rowmap(R, 0, 64), c["enc"], rowmap(R, 0, 64), *c["bn.first_bn1"], p["first_bn1.weight"], p["first_bn1.bias"], ACT_SELU,
                     D.data_ptr() + (Wp + 1) * 64 * es, mp["x_in42"](64), g["first_bn1.weight"], g["first_bn1.bias"], self.bn_ws, self.bn_sums, R, 64)
        # ---- residual blocks, last to first -----------------------------------------------------
        dx0 = None
        for i in range(5, -1, -1):
            ci0, co = FILTS[i]
            ci = _cpad(ci0)
            pre = "encoder.%d.0" % i
            X, Y, D1, D = ws["X%d" % i], ws["Y%d" % i], ws["D1_%d" % i], ws["D%d" % i]
            R1 = B * 43 * W
            d_in = D.data_ptr() + (Wp + 1) * co * es                   # grad wrt block output, interior of D
            # conv2: wgrad, bias grad, dgrad
            self._tn(R, co, 6 * co, d_in, mp["x_in42"](co), Y, mp["y_win42"](co), g[pre + ".conv2.weight"], 6 * co, b_seg=(2, 3 * co, Wp * co),
                      colsum_out=g[pre + ".conv2.bias"])
            wd = self._e(co, 2, 3, co)
            K.copy_strided(p[pre + ".conv2.weight"], wd, 3 * co + 2 * co, (co, 2, 3, co), (1, -3 * co, -co, 6 * co))
            dY = self._e(R1, co)
            ops.gemm_raw(R1, co, 6 * co, D, mp["d_win43"](co), wd, 6 * co, dY, rowmap(R1, 0, co), OCC_F32, self._gd(3 * co, co), a_seg=(2, 3 * co, Wp * co))
            # bn2 + selu backward -> grad wrt conv1 output into D1's interior
            K.bn_act_bwd(dY, rowmap(R1, 0, co), c["o1_%d" % i], rowmap(R1, 0, co), *c["bn2_%d" % i], p[pre + ".bn2.weight"], p[pre + ".bn2.bias"],
                         ACT_SELU, D1.data_ptr() + co * es, mp["y_in43"](co), g[pre + ".bn2.weight"], g[pre + ".bn2.bias"], self.bn_ws,
                         self.bn_sums, R1, co)
            d1_in = D1.data_ptr() + co * es
            self._tn(R1, co, 6 * ci, d1_in, mp["y_in43"](co), X, mp["x_win43"](ci), g[pre + ".conv1.weight"], 6 * ci, b_seg=(2, 3 * ci, Wp * ci),
                      colsum_out=g[pre + ".conv1.bias"])
            # identity path
            if ci0 != co:
                self._tn(R, co, 3 * ci, d_in, mp["x_in42"](co), X.data_ptr() + Wp * ci * es, mp["x_in42"](ci), g[pre + ".conv_downsample.weight"], 3 * ci,
                          colsum_out=g[pre + ".conv_downsample.bias"])
                wdd = self._e(ci, 1, 3, co)
                K.copy_strided(p[pre + ".conv_downsample.weight"], wdd, 2 * ci, (ci, 1, 3, co), (1, 0, -ci, 3 * ci))
                r_addr = self._e(R, ci)
                ops.gemm_raw(R, ci, 3 * co, D.data_ptr() + Wp * co * es, mp["x_in42"](co), wdd, 3 * co, r_addr, rowmap(R, 0, ci), OCC_F32, self._gd(3 * co, co))
                r_map = rowmap(R, 0, ci)
            else:
                r_addr, r_map = d_in, mp["x_in42"](co)
            # conv1 dgrad (+ identity gradient) -> grad wrt block input
            wd1 = self._e(ci, 2, 3, co)
            K.copy_strided(p[pre + ".conv1.weight"], wd1, 3 * ci + 2 * ci, (ci, 2, 3, co), (1, -3 * ci, -ci, 6 * ci))
            if i > 0:
                c_addr, c_map = ws["D%d" % (i - 1)].data_ptr() + (Wp + 1) * ci * es, mp["x_in42"](ci)
            else:
                dx0 = self._e(R, ci)
                c_addr, c_map = dx0, rowmap(R, 0, ci)
            ops.gemm_raw(R, ci, 6 * co, D1, mp["y_win42"](co), wd1, 6 * co, c_addr, c_map, OCC_F32, self._gd(3 * co, co), a_seg=(2, 3 * co, Wp * co),
                         R=r_addr, r_map=r_map, r_dtype=OCC_F32)
        # ---- stem: first_bn + selu, max-pool, LL ---------------------------------------------------
        dpool = self._e(R)
        K.bn_act_bwd(dx0, rowmap(R, 0, 4), c["pool"], rowmap(R, 0, 1), *c["bn.first_bn"], p["first_bn.weight"], p["first_bn.bias"], ACT_SELU,
                     dpool, rowmap(R, 0, 1), g["first_bn.weight"], g["first_bn.bias"], self.bn_ws, self.bn_sums, R, 1)
        dll = K.fill(self._e(B * T, 128), 0.0)
        K.stem_pool_bwd(dpool, c["pool_idx"], dll, B, T, 128)
        feats = c["feats"].view(B * T, -1)
        Fd = feats.shape[1]
        dfe = self._lin_bwd(dll, rowmap(B * T, 0, 128), feats, rowmap(B * T, 0, Fd), B * T, Fd, "LL", 128, need_dx=want_dfeats)
        self.ctx = None
        return dfe.view(B, T, Fd) if want_dfeats else None


def synthetic_backend_params(seed=0):
    """Deterministic stand-in weights in the reference layout (no checkpoint exists offline)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shp in sorted(backend_param_table()):
        leaf = name.rsplit(".", 1)[-1]
        r = torch.randn(shp, generator=g, dtype=torch.float32)
        if len(shp) <= 1 and leaf == "weight":
            t = 1.0 + 0.1 * r
        elif len(shp) <= 1:
            t = 0.05 * r
        elif leaf in ("pos_S", "master1", "master2"):
            t = r
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            if leaf.startswith("att_weight"):
                fan_in = shp[0]
            t = r / math.sqrt(max(fan_in, 1))
        out[name] = t
    return out


class SSLModel(xlsr_mod.SSLModel):
    """sslassist.py:20-49 -- same wrapper as models/xlsr.py but WITHOUT the eval() call (quirk 10)."""
    _eval_at_init = False


class AModel(AliasGuard, torch.nn.Module):
    """Mirror of sslassist.AModel (:432-597): ``AModel(args, device)``; ``forward(x[B,L] or [B,L,1])``.

    Trainable the way the reference trains it (oc_training.py:320-328, 363-385): every back-end tensor is a registered ``nn.Parameter``
    under the reference's name and shape (``LL.weight``, ``encoder.0.0.conv1.weight`` [32,1,2,3], ``GAT_layer_S.att_weight`` ...), a view
    of the engine's flat f32 buffer; with ``finetune_ssl=True`` so are XLS-R's (``ssl_model.model.<fairseq name>``).  A training-mode
    forward is taped through ``autograd_bridge.EngineFunction`` (front-end and back-end each), so ``optim.Adam(aasist.parameters())``,
    ``loss.backward()`` and ``optimizer.step()`` drive the HIP forward / backward kernels; ``occm_amd.trainer.OcTrainer`` is the fused
    fast path over the same engines.  ``nn.DataParallel(aasist)`` works as a pass-through on one visible GPU; multi-GPU data parallelism
    is one process per GPU (``bench.py --gpus N`` / ``occm_amd.parallel``), not replica threads.

    Extra keyword arguments (not in the reference): ``ssl_cfg`` / ``ssl_dtype`` / ``ssl_state_dict`` choose the
    XLS-R variant, compute dtype and weights; ``backend_state_dict`` the AASIST weights; both fall back to the
    deterministic synthetic filler because no checkpoint exists offline (the reference hard-codes a path, :24)."""

    def __init__(self, args=None, device="cuda", ssl_cfg=None, ssl_dtype=torch.bfloat16, ssl_state_dict=None, backend_state_dict=None, seed=0,
                 backend_compute=None, finetune_ssl=False, ssl_cp_path=None, synthetic_ssl=False, ssl_train_cfg=None, ssl_model=None, ssl_f32_gemm="exact"):
        """ssl_model: a ready front-end module offering ``extract_feat(x) -> f32 [B,T,1024]`` to use instead of building ``SSLModel``
        (the parity tests pass a stub that hands seeded features through, exactly as the golden generator stubs the reference's)."""
        super().__init__()
        self.device = device
        self.ssl_model = ssl_model if ssl_model is not None else \
            SSLModel(device, cp_path=ssl_cp_path, state_dict=ssl_state_dict, cfg=ssl_cfg, dtype=ssl_dtype, seed=seed, finetune=finetune_ssl,
                     synthetic=synthetic_ssl, train_cfg=ssl_train_cfg, f32_gemm=ssl_f32_gemm)
        if backend_compute is None:
            backend_compute = "bf16" if ssl_dtype == torch.bfloat16 else "f32"
        self.backend = AasistBackend(backend_state_dict, device=device, seed=seed, compute=backend_compute)
        self.param_set = attach_parameters(self, self.backend)
        self.dropout_masks = None      # None: draw masks on the device; {}: every back-end dropout off (the p = 0 of a parity run)

    supports_lengths = True

    def forward(self, x, masks=None, lengths=None):
        """lengths (evaluation only, not in the reference): sample counts of a ZERO-PADDED batch of unequal utterances.  The front-end runs the
        whole batch with key masks (``XlsrFrontend.forward``); the back-end -- whose pooling, graph sizes and top-k counts depend on the
        frame count -- runs once per distinct frame count on the un-padded rows.  Row b of the result equals ``forward(x[b:b+1, :lengths[b]])``."""
        x = x.squeeze(-1) if x.dim() == 3 else x
        if lengths is not None:
            if self.training:
                raise OccError("lengths (masked batches) are an evaluation feature: call model.eval() first")
            return self._forward_lengths(x, [int(v) for v in lengths])
        feats = self.ssl_model.extract_feat(x)
        masks = self.dropout_masks if masks is None else masks
        be = self.backend
        if not self.training:
            with torch.no_grad():
                return be.forward(feats, train=False, masks=masks)

        def bwd(grads, needs):
            demb, dlog = (None if g is None else g.contiguous().float() for g in grads)
            if dlog is None:
                dlog = torch.zeros(feats.shape[0], 2, device=be.device)
            d = be.backward(demb, dlog, want_dfeats=bool(needs[0]))
            return (d,)

        return run_engine(self.param_set, lambda f: be.forward(f, train=True, masks=masks), bwd, feats)

    def _forward_lengths(self, x, lengths):
        with torch.no_grad():
            feats = self.ssl_model.extract_feat(x, lengths=lengths)
            fr = [xlsr_mod.n_frames(v) for v in lengths]
            emb = torch.empty(len(fr), 160, device=feats.device); out = torch.empty(len(fr), 2, device=feats.device)
            for T in sorted(set(fr)):
                idx = torch.tensor([i for i, t in enumerate(fr) if t == T], device=feats.device)
                e, o = self.backend.forward(feats[idx, :T].contiguous(), train=False)
                emb[idx] = e; out[idx] = o
        return emb, out

    def backward(self, demb, dlogits):
        return self.backend.backward(demb, dlogits)

    def state_dict(self, *a, **kw):
        """Reference key set (oc_training.py:401 saves ``aasist.module.state_dict()``): the back-end's 247 keys plus EVERY
        ``ssl_model.model.*`` tensor of the loaded fairseq checkpoint -- also the ones the features_only forward never reads --
        so the reference's strict ``load_state_dict`` (oc_classifier.py:340) accepts the file."""
        sd = self.backend.state_dict()
        full = self.ssl_model.full_state_dict().items() if hasattr(self.ssl_model, "full_state_dict") else ()
        for k, v in full:
            sd["ssl_model.model." + k] = v.detach().clone() if torch.is_tensor(v) else v
        return sd

    def load_state_dict(self, sd, strict=True):
        """strict (the reference's default): every back-end key and every ``ssl_model.model.*`` tensor of the path must be present
        with the right shape -- a checkpoint can never leave random tensors in place silently."""
        self.backend.load_reference_params(sd, strict=strict)
        ssl = {k[len("ssl_model.model."):]: v for k, v in sd.items() if k.startswith("ssl_model.model.")}
        if (ssl or strict) and hasattr(self.ssl_model, "load_params"):
            self.ssl_model.load_params(ssl, strict=strict)
        return self
