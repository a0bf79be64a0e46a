"""LCNN back-end on MI355X -- drop-in for ``models/lcnn.py`` (mfm :121-136, group :139-150, LCNN :152-217, lcnn_net :228-230,
ssl_lcnn :233-263), in the ``asoftmax=False`` form the reference builds (lcnn.py:244, occm.py:52).

``lcnn_net()`` returns a module whose ``forward(x[B,1,T,1024]) -> logits [B,2]`` and state_dict keys match the reference (including the
``group.bn`` BatchNorm that the reference constructs and never applies: it is carried through load / save untouched).  Forward and
backward are explicit, f32 channels-last: the 5x5 / 3x3 / 1x1 convolutions are implicit GEMMs over zero-bordered buffers (occ_gemm,
occ_gemm_tn for the weight gradients, a tap-reversed occ_gemm for the input gradients), Max-Feature-Map is fused with the 2x2 max-pool
that follows every conv group (occ_mfm_pool2_*), BatchNorm / dropout / adaptive pooling are the kernels of csrc/backend.hip and
csrc/lcnn.hip.
"""
import ctypes
import math

import torch

from .. import backend_ops as K
from .. import ops
from .._lib import ACT_NONE, OCC_F32, OCC_F32_AS_BF16, OccError, check, lib, ptr, require_gpu, stream_ptr
from ..autograd_bridge import AliasGuard, attach_parameters, run_engine
from ..ops import rowmap

C_S = [128, 64, 32, 16, 8, 4, 2]            # lcnn.py:153
P_FC = {"fc0": 0.75, "fc1": 0.75, "fc2": 0.0}   # Dropout inside the three linear mfm blocks (lcnn.py:172-181)
POOL_W = 64                                 # AdaptiveAvgPool2d((1, 64)), lcnn.py:169


def _rm(m):
    return ctypes.byref(m)


def _cp(c):
    return 4 if c < 4 else c


def lcnn_param_table():
    """(name, reference shape, kind) in the reference's state_dict order; kind: conv / vec / fc / buf."""
    t = []

    def conv(pre, ci, co, k):
        t.append((pre + ".filter.weight", (2 * co, ci, k, k), "conv")); t.append((pre + ".filter.bias", (2 * co,), "vec"))

    def bn(pre, c):
        t.append((pre + ".weight", (c,), "vec")); t.append((pre + ".bias", (c,), "vec"))
        t.append((pre + ".running_mean", (c,), "buf")); t.append((pre + ".running_var", (c,), "buf")); t.append((pre + ".num_batches_tracked", (), "buf"))

    def grp(pre, ci, co):
        conv(pre + ".conv_a", ci, ci, 1); bn(pre + ".bn", ci); conv(pre + ".conv", ci, co, 3)

    def fc(pre, i, o):
        t.append((pre + ".filter.0.weight", (2 * o, i), "fc")); t.append((pre + ".filter.0.bias", (2 * o,), "vec"))

    conv("layer1.0", 1, C_S[5], 5)
    grp("layer2.0", C_S[5], C_S[4]); bn("layer2.2", C_S[4])
    grp("layer3.0", C_S[4], C_S[3]); bn("layer3.2", C_S[3])
    fc("fc0.0", C_S[3] * POOL_W, 32); fc("fc1.0", 32, 32); fc("fc2.0", 32, 8)
    t.append(("fc3.weight", (2, 8), "fc")); t.append(("fc3.bias", (2,), "vec"))
    return t


class LcnnBackend:
    graph_safe = True           # a train step makes no host decision and draws its masks from device-side counters: OcTrainer may replay it from a HIP graph

    def __init__(self, params=None, device="cuda", seed=4, compute="f32"):
        """compute: "f32" (exact-f32 MFMA, the parity path) or "bf16" (operands rounded to bf16 on the way into LDS wherever K and the
        operand strides allow, f32 accumulate); activations, parameters and gradients are f32 in memory either way."""
        require_gpu()
        if compute not in ("f32", "bf16"):
            raise OccError("compute must be 'f32' or 'bf16'")
        self.compute = compute
        self.device = torch.device(device)
        self.table = lcnn_param_table()
        self.slots, off = {}, 0
        for name, shp, kind in self.table:
            if kind == "buf":
                continue
            ishp = (shp[0], shp[2], shp[3], _cp(shp[1])) if kind == "conv" else tuple(shp)
            n = 1
            for d in ishp:
                n *= d
            self.slots[name] = (off, ishp, n)
            off += (n + 3) // 4 * 4
        self.P = torch.zeros(off, device=self.device)
        self.G = torch.zeros(off, device=self.device)
        self.p = {k: self.P[o:o + n].view(s) for k, (o, s, n) in self.slots.items()}
        self.g = {k: self.G[o:o + n].view(s) for k, (o, s, n) in self.slots.items()}
        self.buf = {}
        for name, shp, kind in self.table:
            if kind == "buf":
                self.buf[name] = torch.zeros(1, device=self.device, dtype=torch.int64) if name.endswith("num_batches_tracked") else \
                    (torch.ones(shp, device=self.device) if name.endswith("running_var") else torch.zeros(shp, device=self.device))
        self.bn_ws = torch.empty(512 * 256 * 2, device=self.device, dtype=torch.float64)
        self.bn_sums = torch.empty(512, device=self.device)
        self._ws, self.ctx = {}, None
        self.rng_seed, self.rng_step = seed, 0
        self.rng_step_dev = torch.zeros(1, device=self.device, dtype=torch.int64)      # the step counter the device-drawn masks are keyed by
        self.load_reference_params(params if params is not None else synthetic_lcnn_params(seed))

    # ---------------------------------------------------------------------------- checkpoint layouts --
    def load_reference_params(self, sd):
        with torch.no_grad():
            for name, shp, kind in self.table:
                if name not in sd:
                    raise OccError("LCNN state_dict lacks %s" % name)
                v = sd[name].detach().to(self.device)
                if tuple(v.shape) != tuple(shp):
                    raise OccError("shape of %s is %s, expected %s" % (name, tuple(v.shape), tuple(shp)))
                if kind == "buf":
                    self.buf[name].copy_(v.reshape(self.buf[name].shape))
                elif kind == "conv":
                    self.p[name].zero_(); self.p[name][..., : shp[1]].copy_(v.float().permute(0, 2, 3, 1))
                else:
                    self.p[name].copy_(v.float().reshape(self.p[name].shape))

    def ref_views(self, flat):
        """{reference name: VIEW of ``flat`` (self.P / self.G / same-sized buffer) in the reference's shape}."""
        out = {}
        for name, shp, kind in self.table:
            if kind == "buf":
                continue
            o, ishp, n = self.slots[name]
            v = flat[o:o + n].view(ishp)
            out[name] = v[..., : shp[1]].permute(0, 3, 1, 2) if kind == "conv" else v.view(shp)
        return out

    def _export(self, src):
        out = {}
        for name, shp, kind in self.table:
            if kind == "conv":
                out[name] = src[name][..., : shp[1]].permute(0, 3, 1, 2).contiguous().clone()
            elif kind != "buf":
                out[name] = src[name].detach().clone().reshape(shp)
        return out

    def state_dict(self):
        exp = self._export(self.p)
        out = {}
        for name, shp, kind in self.table:             # the reference's key order
            if kind == "buf":
                v = self.buf[name]
                out[name] = v.clone().reshape(()) if name.endswith("num_batches_tracked") else v.clone()
            else:
                out[name] = exp[name]
        return out

    def grad_dict(self):
        return self._export(self.g)

    def zero_grad(self):
        K.fill(self.G, 0.0)

    # --------------------------------------------------------------------------------------- helpers --
    def _gemm(self, M, N, Kd, A, a_map, W, ldw, C, c_map, a_seg=None, **kw):
        ab = OCC_F32
        if self.compute == "bf16" and Kd % 8 == 0 and ldw % 8 == 0 and a_map.row_stride % 8 == 0 and a_map.batch_stride % 8 == 0 and \
                a_map.line_stride % 8 == 0 and (a_seg is None or (a_seg[1] % 8 == 0 and a_seg[2] % 8 == 0)):
            ab = OCC_F32_AS_BF16
        return ops.gemm_raw(M, N, Kd, A, a_map, W, ldw, C, c_map, OCC_F32, ab, a_seg=a_seg, **kw)

    def _tn(self, *a, **kw):
        return K.gemm_tn(*a, bf16_mfma=(self.compute == "bf16"), **kw)

    def _e(self, *s, dtype=torch.float32):
        return torch.empty(*s, device=self.device, dtype=dtype)

    def _z(self, *s):
        return torch.zeros(*s, device=self.device)

    def _flipk(self, w, ci, co, k):
        """[co,k,k,ci] -> [ci,k,k,co] with both taps reversed: the operand of the input-gradient correlation of a stride-1 'same' conv."""
        out = self._e(ci, k, k, co)
        K.copy_strided(w, out, (k - 1) * k * ci + (k - 1) * ci, (ci, k, k, co), (1, -k * ci, -ci, k * k * ci))
        return out

    def _geometry(self, B, H0, W0):
        key = (B, H0, W0)
        if key in self._ws:
            self._ws[key] = self._ws.pop(key)
            return self._ws[key]
        H1, W1 = H0 // 2, W0 // 2
        H2, W2 = H1 // 2, W1 // 2
        H3, W3 = H2 // 2, W2 // 2
        if H3 < 1 or W3 < 1:
            raise OccError("LCNN input [%d, %d] is too small for three 2x2 poolings" % (H0, W0))
        c4, c8, c16 = C_S[5], C_S[4], C_S[3]
        g = {"hw": [(H0, W0), (H1, W1), (H2, W2), (H3, W3)],
             "X0": self._z(B, H0 + 4, W0 + 4, 4),                 # the image in channel 0, 2-pixel zero border (5x5, padding 2)
             "Y2": self._z(B, H1 + 2, W1 + 2, c4),                # mfm(conv_a) of layer2 with a 1-pixel border
             "Y3": self._z(B, H2 + 2, W2 + 2, c8),
             "D1": self._z(B, H0 + 4, W0 + 4, 2 * c4),            # gradients wrt the conv outputs, bordered for the input-gradient correlation
             "D2": self._z(B, H1 + 2, W1 + 2, 2 * c8),
             "D3": self._z(B, H2 + 2, W2 + 2, 2 * c16)}
        self._ws[key] = g
        while len(self._ws) > 4:
            self._ws.pop(next(iter(self._ws)))
        return g

    def _bn_fwd(self, name, x, rows, C, y, train, c):
        mean, rstd = self._e(C), self._e(C)
        K.bn_stats(x, rowmap(rows, 0, C), rows, C, self.bn_ws, mean, rstd, self.buf[name + ".running_mean"], self.buf[name + ".running_var"],
                   self.buf[name + ".num_batches_tracked"], train)
        K.bn_act_fwd(x, rowmap(rows, 0, C), mean, rstd, self.p[name + ".weight"], self.p[name + ".bias"], ACT_NONE, y, rowmap(rows, 0, C), rows, C)
        c[name] = (mean, rstd)

    def _bn_bwd(self, name, dy, x, rows, C, dx, c):
        mean, rstd = c[name]
        full = rowmap(rows, 0, C)
        K.bn_act_bwd(dy, full, x, full, mean, rstd, self.p[name + ".weight"], self.p[name + ".bias"], ACT_NONE, dx, full,
                     self.g[name + ".weight"], self.g[name + ".bias"], self.bn_ws, self.bn_sums, rows, C)

    def _fc_fwd(self, site, x, B, i, o2, train, masks, c):
        """mfm(type=0): Linear(i, o2) -> Dropout -> max of the halves.  Keeps the dropped-out pre-MFM values and the mask."""
        pre = site + ".0.filter.0"
        z = self._e(B, o2)
        self._gemm(B, o2, i, x, rowmap(B, 0, i), self.p[pre + ".weight"], i, z, rowmap(B, 0, o2), bias=self.p[pre + ".bias"])
        mask, pd = None, P_FC[site]
        if train and pd > 0:
            if masks is not None:
                mask = masks[site].to(self.device, torch.uint8).contiguous() if site in masks else None
                if mask is not None:
                    K.dropout(z, z, mask, pd, 0, 0, generate=False)
            else:
                mask = torch.empty(B, o2, device=self.device, dtype=torch.uint8)
                self._site_id += 1
                K.dropout_step(z, z, mask, pd, self.rng_seed, self.rng_step_dev, self._site_id)
        h = self._e(B, o2 // 2)
        check(lib().occ_mfm_fwd(ptr(z), ptr(h), _rm(rowmap(B, 0, o2 // 2)), B, o2 // 2, stream_ptr()), "occ_mfm_fwd")
        c[site] = (x, z, mask)
        return h

    def _fc_bwd(self, site, dh, B, i, o2, c, need_dx=True):
        pre = site + ".0.filter.0"
        x, z, mask = c[site]
        dz = self._e(B, o2)
        check(lib().occ_mfm_bwd(ptr(dh), _rm(rowmap(B, 0, o2 // 2)), ptr(z), ptr(dz), B, o2 // 2, stream_ptr()), "occ_mfm_bwd")
        if mask is not None:
            K.dropout(dz, dz, mask, P_FC[site], 0, 0, generate=False)
        self._tn(B, o2, i, dz, rowmap(B, 0, o2), x, rowmap(B, 0, i), self.g[pre + ".weight"], i, colsum_out=self.g[pre + ".bias"])
        if not need_dx:
            return None
        wt = self._e(i, o2)
        K.copy_strided(self.p[pre + ".weight"], wt, 0, (1, 1, i, o2), (0, 0, 1, i))
        dx = self._e(B, i)
        self._gemm(B, i, o2, dz, rowmap(B, 0, o2), wt, o2, dx, rowmap(B, 0, i))
        return dx

    # ======================================================================================= forward ==
    def forward(self, x, train=False, masks=None):
        """x f32 [B,1,T,D] (or [B,T,D] front-end features), D = 1024 for the reference's fc0 -> logits [B,2].
        masks: None -> dropout masks drawn on the device (train only); dict -> injected keep-masks {"fc0","fc1"} (absent = no dropout)."""
        x = x.to(self.device, torch.float32)
        if x.dim() == 3:
            x = x.unsqueeze(1)
        if x.dim() != 4 or x.shape[1] != 1:
            raise OccError("lcnn expects [B,1,T,D]")
        B, _, H0, W0 = x.shape
        geo = self._geometry(B, H0, W0)
        (_, _), (H1, W1), (H2, W2), (H3, W3) = geo["hw"]
        c4, c8, c16 = C_S[5], C_S[4], C_S[3]
        if c16 * POOL_W != self.p["fc0.0.filter.0.weight"].shape[1]:
            raise OccError("fc0 expects %d inputs" % (c16 * POOL_W))
        c = {"B": B, "geo": geo, "train": train}
        self._site_id = 0
        p, es = self.p, 4
        X0, Wp0 = geo["X0"], W0 + 4
        K.copy_rows(x.contiguous().view(-1), rowmap(B * H0 * W0, 0, 1), X0.data_ptr() + (2 * Wp0 + 2) * 4 * es, rowmap(H0 * W0, (H0 + 4) * Wp0 * 4, 4, W0, Wp0 * 4),
                    B * H0 * W0, 1)
        # ---- layer1: mfm(1, 4, 5, 1, 2) + MaxPool2d(2, 2) ------------------------------------------------------------------------
        R0, R1, R2, R3 = B * H0 * W0, B * H1 * W1, B * H2 * W2, B * H3 * W3
        o1 = self._e(R0, 2 * c4)
        self._gemm(R0, 2 * c4, 100, X0, rowmap(H0 * W0, (H0 + 4) * Wp0 * 4, 4, W0, Wp0 * 4), p["layer1.0.filter.weight"], 100, o1, rowmap(R0, 0, 2 * c4),
                   a_seg=(5, 20, Wp0 * 4), bias=p["layer1.0.filter.bias"])
        p1, i1 = self._e(R1, c4), torch.empty(R1 * c4, device=self.device, dtype=torch.uint8)
        check(lib().occ_mfm_pool2_fwd(ptr(o1), ptr(p1), _rm(rowmap(R1, 0, c4)), ptr(i1), B, H0, W0, c4, stream_ptr()), "occ_mfm_pool2_fwd")
        if not train:
            del o1

        # ---- layer2 / layer3: group (1x1 mfm, 3x3 mfm) + MaxPool2d + BatchNorm2d ------------------------------------------------------
        def group(tag, xin, R, H, W, ci, co, Y, bn_name):
            Wp = W + 2
            a = self._e(R, 2 * ci)
            self._gemm(R, 2 * ci, _cp(ci), xin, rowmap(R, 0, ci), p[tag + ".conv_a.filter.weight"], _cp(ci), a, rowmap(R, 0, 2 * ci), bias=p[tag + ".conv_a.filter.bias"])
            y_in = rowmap(H * W, (H + 2) * Wp * ci, ci, W, Wp * ci)
            check(lib().occ_mfm_fwd(ptr(a), Y.data_ptr() + (Wp + 1) * ci * es, _rm(y_in), R, ci, stream_ptr()), "occ_mfm_fwd")
            cv = self._e(R, 2 * co)
            self._gemm(R, 2 * co, 9 * ci, Y, y_in, p[tag + ".conv.filter.weight"], 9 * ci, cv, rowmap(R, 0, 2 * co), a_seg=(3, 3 * ci, Wp * ci),
                       bias=p[tag + ".conv.filter.bias"])
            Ro = B * (H // 2) * (W // 2)
            pl, idx = self._e(Ro, co), torch.empty(Ro * co, device=self.device, dtype=torch.uint8)
            check(lib().occ_mfm_pool2_fwd(ptr(cv), ptr(pl), _rm(rowmap(Ro, 0, co)), ptr(idx), B, H, W, co, stream_ptr()), "occ_mfm_pool2_fwd")
            bo = self._e(Ro, co)
            self._bn_fwd(bn_name, pl, Ro, co, bo, train, c)
            c[tag] = (xin, a, pl, idx)
            return bo

        b2 = group("layer2.0", p1, R1, H1, W1, c4, c8, geo["Y2"], "layer2.2")
        b3 = group("layer3.0", b2, R2, H2, W2, c8, c16, geo["Y3"], "layer3.2")
        # ---- AdaptiveAvgPool2d((1, 64)) + flatten, the three linear mfm blocks, fc3 -------------------------------------------------
        f = self._e(B, c16 * POOL_W)
        check(lib().occ_adaptive_avgpool_1xw_fwd(ptr(b3), ptr(f), B, H3, W3, c16, POOL_W, stream_ptr()), "occ_adaptive_avgpool_1xw_fwd")
        h0 = self._fc_fwd("fc0", f, B, c16 * POOL_W, 64, train, masks, c)
        h1 = self._fc_fwd("fc1", h0, B, 32, 64, train, masks, c)
        h2 = self._fc_fwd("fc2", h1, B, 32, 16, train, masks, c)
        self._w3 = self._fc3_pad()
        out = self._e(B, 4)
        self._gemm(B, 4, 8, h2, rowmap(B, 0, 8), self._w3[0], 8, out, rowmap(B, 0, 4), bias=self._w3[1])
        c["i1"], c["h2"] = i1, h2
        self.ctx = c if train else None
        if train:
            self.rng_step += 1
            K.add_u64(self.rng_step_dev, 1)
        return out[:, :2].contiguous()

    def _fc3_pad(self):
        """fc3 has 2 outputs; the GEMM wants N % 4 == 0 -> zero-padded copy [4,8] (+bias [4])."""
        w, b = self._z(4, 8), self._z(4)
        w[:2].copy_(self.p["fc3.weight"]); b[:2].copy_(self.p["fc3.bias"])
        return w, b

    # ====================================================================================== backward ==
    def backward(self, dlogits, want_dfeats=False):
        """Accumulates parameter gradients into G; with want_dfeats returns d loss / d input [B,T,D]."""
        c = self.ctx
        if c is None:
            raise OccError("backward() needs a preceding forward(train=True)")
        B, geo, p, g, es = c["B"], c["geo"], self.p, self.g, 4
        (H0, W0), (H1, W1), (H2, W2), (H3, W3) = geo["hw"]
        c4, c8, c16 = C_S[5], C_S[4], C_S[3]
        R0, R1, R2, R3 = B * H0 * W0, B * H1 * W1, B * H2 * W2, B * H3 * W3
        # ---- fc3 and the linear mfm blocks ---------------------------------------------------------------------------------------
        dl4 = self._z(B, 4); dl4[:, :2].copy_(dlogits)
        gw4, gb4 = self._z(4, 8), self._z(4)
        self._tn(B, 4, 8, dl4, rowmap(B, 0, 4), c["h2"], rowmap(B, 0, 8), gw4, 8, colsum_out=gb4)
        K.axpby(gw4[:2].contiguous().view(-1), g["fc3.weight"].view(-1), g["fc3.weight"].view(-1))
        K.axpby(gb4[:2].contiguous(), g["fc3.bias"], g["fc3.bias"])
        wt3 = self._e(8, 4)
        K.copy_strided(self._w3[0], wt3, 0, (1, 1, 8, 4), (0, 0, 1, 8))
        dh2 = self._e(B, 8)
        self._gemm(B, 8, 4, dl4, rowmap(B, 0, 4), wt3, 4, dh2, rowmap(B, 0, 8))
        dh1 = self._fc_bwd("fc2", dh2, B, 32, 16, c)
        dh0 = self._fc_bwd("fc1", dh1, B, 32, 64, c)
        df = self._fc_bwd("fc0", dh0, B, c16 * POOL_W, 64, c)
        db3 = self._e(R3, c16)
        check(lib().occ_adaptive_avgpool_1xw_bwd(ptr(df), ptr(db3), B, H3, W3, c16, POOL_W, stream_ptr()), "occ_adaptive_avgpool_1xw_bwd")

        # ---- conv groups, last first ------------------------------------------------------------------------------------------------
        def group_bwd(tag, dbo, R, H, W, ci, co, Y, D, bn_name):
            """dbo: gradient wrt the BatchNorm output [Ro, co]; returns the gradient wrt the group's input [R, ci]."""
            xin, a, pl, idx = c[tag]
            Wp = W + 2
            Ro = B * (H // 2) * (W // 2)
            dpl = self._e(Ro, co)
            self._bn_bwd(bn_name, dbo, pl, Ro, co, dpl, c)
            d_in = rowmap(H * W, (H + 2) * Wp * 2 * co, 2 * co, W, Wp * 2 * co)          # interior of D [B,H+2,W+2,2co]
            d_addr = D.data_ptr() + (Wp + 1) * 2 * co * es
            check(lib().occ_mfm_pool2_bwd(ptr(dpl), _rm(rowmap(Ro, 0, co)), ptr(idx), d_addr, _rm(d_in), B, H, W, co, stream_ptr()), "occ_mfm_pool2_bwd")
            y_in = rowmap(H * W, (H + 2) * Wp * ci, ci, W, Wp * ci)
            self._tn(R, 2 * co, 9 * ci, d_addr, d_in, Y, y_in, g[tag + ".conv.filter.weight"], 9 * ci, b_seg=(3, 3 * ci, Wp * ci),
                     colsum_out=g[tag + ".conv.filter.bias"])
            wf = self._flipk(p[tag + ".conv.filter.weight"], ci, 2 * co, 3)
            dy = self._e(R, ci)                                                          # gradient wrt mfm(conv_a)
            self._gemm(R, ci, 9 * 2 * co, D, d_in, wf, 9 * 2 * co, dy, rowmap(R, 0, ci), a_seg=(3, 3 * 2 * co, Wp * 2 * co))
            da = self._e(R, 2 * ci)
            check(lib().occ_mfm_bwd(ptr(dy), _rm(rowmap(R, 0, ci)), ptr(a), ptr(da), R, ci, stream_ptr()), "occ_mfm_bwd")
            self._tn(R, 2 * ci, _cp(ci), da, rowmap(R, 0, 2 * ci), xin, rowmap(R, 0, ci), g[tag + ".conv_a.filter.weight"], _cp(ci),
                     colsum_out=g[tag + ".conv_a.filter.bias"])
            wt = self._e(_cp(ci), 2 * ci)
            K.copy_strided(p[tag + ".conv_a.filter.weight"], wt, 0, (1, 1, _cp(ci), 2 * ci), (0, 0, 1, _cp(ci)))
            dx = self._e(R, ci)
            self._gemm(R, ci, 2 * ci, da, rowmap(R, 0, 2 * ci), wt, 2 * ci, dx, rowmap(R, 0, ci))
            return dx

        db2 = group_bwd("layer3.0", db3, R2, H2, W2, c8, c16, geo["Y3"], geo["D3"], "layer3.2")
        dp1 = group_bwd("layer2.0", db2, R1, H1, W1, c4, c8, geo["Y2"], geo["D2"], "layer2.2")
        # ---- layer1 -------------------------------------------------------------------------------------------------------------------
        Wp0 = W0 + 4
        D1 = geo["D1"]
        d1_in = rowmap(H0 * W0, (H0 + 4) * Wp0 * 2 * c4, 2 * c4, W0, Wp0 * 2 * c4)
        d1_addr = D1.data_ptr() + (2 * Wp0 + 2) * 2 * c4 * es
        check(lib().occ_mfm_pool2_bwd(ptr(dp1), _rm(rowmap(R1, 0, c4)), ptr(c["i1"]), d1_addr, _rm(d1_in), B, H0, W0, c4, stream_ptr()), "occ_mfm_pool2_bwd")
        x_win = rowmap(H0 * W0, (H0 + 4) * Wp0 * 4, 4, W0, Wp0 * 4)
        self._tn(R0, 2 * c4, 100, d1_addr, d1_in, geo["X0"], x_win, g["layer1.0.filter.weight"], 100, b_seg=(5, 20, Wp0 * 4), colsum_out=g["layer1.0.filter.bias"])
        self.ctx = None
        if not want_dfeats:
            return None
        wf = self._flipk(p["layer1.0.filter.weight"], 4, 2 * c4, 5)                       # [4 (channel 0 real), 5, 5, 8]
        dx4 = self._e(R0, 4)
        self._gemm(R0, 4, 25 * 2 * c4, D1, d1_in, wf, 25 * 2 * c4, dx4, rowmap(R0, 0, 4), a_seg=(5, 5 * 2 * c4, Wp0 * 2 * c4))
        dx = self._e(B, H0, W0)
        K.copy_rows(dx4, rowmap(R0, 0, 4), dx, rowmap(R0, 0, 1), R0, 1)
        return dx


def synthetic_lcnn_params(seed=4):
    """Random parameters with sane magnitudes (tests, bench); NOT the reference's init_weight."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shp, kind in sorted(lcnn_param_table()):
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[name] = torch.zeros((), dtype=torch.int64)
            continue
        r = torch.randn(shp, generator=g)
        if leaf == "running_var":
            out[name] = 0.5 + r.abs()
        elif leaf == "running_mean":
            out[name] = 0.1 * r
        elif len(shp) <= 1 and leaf == "weight":
            out[name] = 1.0 + 0.1 * r
        elif len(shp) <= 1:
            out[name] = 0.05 * r
        else:
            fan = 1
            for d in shp[1:]:
                fan *= d
            out[name] = r / math.sqrt(fan)
    return out


class _Lcnn(AliasGuard, torch.nn.Module):
    """Module facade with the reference's call signature (lcnn.py:186: ``forward(x, eval=False)``); parameters and autograd as
    ``models.senet._SeResNet``."""

    def __init__(self, state_dict=None, device="cuda", seed=4, compute="f32", asoftmax=False, **kwargs):
        super().__init__()
        if asoftmax:
            raise OccError("the A-softmax head (lcnn.py:22-119) is not on the path: the reference builds lcnn_net(asoftmax=False) only")
        self.backend = LcnnBackend(state_dict, device=device, seed=seed, compute=compute)
        self.param_set = attach_parameters(self, self.backend)

    def forward(self, x, eval=False, masks=None):
        be = self.backend
        if not self.training:
            with torch.no_grad():
                return be.forward(x, train=False, masks=masks)

        def bwd(grads, needs):
            d = be.backward(grads[0].contiguous().float(), want_dfeats=bool(needs[0]))
            return (d.view(x.shape) if needs[0] else None,)

        return run_engine(self.param_set, lambda t: be.forward(t, train=True, masks=masks), bwd, x)

    def backward(self, dlogits, want_dfeats=False):
        return self.backend.backward(dlogits, want_dfeats=want_dfeats)

    def state_dict(self, *a, **kw):
        return self.backend.state_dict()

    def load_state_dict(self, sd, strict=True):
        self.backend.load_reference_params(sd)
        return self

    def cuda(self, *a, **kw):
        return self

    def to(self, *a, **kw):
        return self


def lcnn_net(**kwargs):
    """lcnn.py:228-230."""
    return _Lcnn(**kwargs)


class ssl_lcnn(torch.nn.Module):
    """lcnn.py:233-263: XLS-R features [B,T,1024] -> unsqueeze(1) -> LCNN logits [B,2]."""

    def __init__(self, device="cuda", ssl_cfg=None, ssl_dtype=torch.bfloat16, ssl_state_dict=None, state_dict=None, finetune_ssl=False, seed=4,
                 backend_compute=None, ssl_cp_path=None, synthetic_ssl=False):
        super().__init__()
        from .xlsr import SSLModel
        self.frontend = SSLModel(device, cp_path=ssl_cp_path, state_dict=ssl_state_dict, cfg=ssl_cfg, dtype=ssl_dtype, finetune=finetune_ssl,
                                 synthetic=synthetic_ssl)
        if backend_compute is None:
            backend_compute = "bf16" if ssl_dtype == torch.bfloat16 else "f32"
        self.lcnn = lcnn_net(state_dict=state_dict, device=device, seed=seed, compute=backend_compute)
        self.ssl_model = self.frontend
        self.backend = self.lcnn.backend

    def forward(self, x):
        feats = self.frontend.extract_feat(x)
        self.lcnn.train(self.training)
        return self.lcnn(feats.unsqueeze(1))

    def state_dict(self, *a, **kw):
        """Keys as the reference module tree gives them: ``lcnn.*`` and ``frontend.model.*`` (fairseq names)."""
        sd = {"lcnn." + k: v for k, v in self.lcnn.state_dict().items()}
        for k, v in self.frontend.full_state_dict().items():
            sd["frontend.model." + k] = v.detach().clone() if torch.is_tensor(v) else v
        return sd

    def load_state_dict(self, sd, strict=True):
        self.lcnn.load_state_dict({k[len("lcnn."):]: v for k, v in sd.items() if k.startswith("lcnn.")})
        ssl = {k[len("frontend.model."):]: v for k, v in sd.items() if k.startswith("frontend.model.")}
        if ssl or strict:
            self.frontend.load_params(ssl, strict=strict)
        return self
