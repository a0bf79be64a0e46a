"""SE-ResNet34 back-end on MI355X -- drop-in for ``models/senet.py`` (SELayer :13-28, SEBasicBlock :31-61,
ResNet :64-152, se_resnet34 :154-156, ssl_resnet34 :162-185).

``se_resnet34()`` returns a module whose ``forward(x[B,1,T,D]) -> (com[B,128], des[B,2])`` and state_dict keys match the
reference.  Forward and backward are explicit (no autograd tape), f32 channels-last, every convolution an implicit GEMM
over zero-bordered buffers (occ_gemm / occ_gemm_tn), BatchNorm+ReLU, max-pool, squeeze-excite and the pools are the HIP
kernels of csrc/backend.hip.  Strided 3x3 convolutions back-propagate as four parity-class GEMMs (even/odd row x even/odd
column of the input each see a fixed subset of the taps), the 1x1 stride-2 shortcut as one more on the even/even class.
"""
import ctypes
import math

import torch

from .. import backend_ops as K
from .. import ops
from .._lib import ACT_NONE, ACT_RELU, OCC_F32, OCC_F32_AS_BF16, OccError, check, lib, ptr, require_gpu, stream_ptr
from ..autograd_bridge import AliasGuard, attach_parameters, run_engine
from ..ops import rowmap

LAYERS = [3, 4, 6, 3]                       # senet.py:154-156
CHANNELS = [16, 16, 32, 64, 128]            # senet.py:66


def _rm(m):
    return ctypes.byref(m)


def senet_param_table():
    t = []

    def bn(pre, c):
        t.append((pre + ".weight", (c,))); t.append((pre + ".bias", (c,)))

    t.append(("conv1.weight", (CHANNELS[0], 1, 7, 7))); bn("bn1", CHANNELS[0])
    inpl = CHANNELS[0]
    for li, nb in enumerate(LAYERS):
        p = CHANNELS[li + 1]
        for bi in range(nb):
            pre = "layer%d.%d" % (li + 1, bi)
            stride = 2 if (li > 0 and bi == 0) else 1
            t.append((pre + ".conv1.weight", (p, inpl, 3, 3))); bn(pre + ".bn1", p)
            t.append((pre + ".conv2.weight", (p, p, 3, 3))); bn(pre + ".bn2", p)
            t.append((pre + ".se.fc.0.weight", (p // 16, p))); t.append((pre + ".se.fc.2.weight", (p, p // 16)))
            if stride != 1 or inpl != p:
                t.append((pre + ".downsample.0.weight", (p, inpl, 1, 1))); bn(pre + ".downsample.1", p)
            inpl = p
    t += [("embedding.weight", (128, 128)), ("embedding.bias", (128,)), ("classifier.weight", (2, 128)), ("classifier.bias", (2,))]
    return t


def _blocks():
    out, inpl = [], CHANNELS[0]
    for li, nb in enumerate(LAYERS):
        p = CHANNELS[li + 1]
        for bi in range(nb):
            stride = 2 if (li > 0 and bi == 0) else 1
            out.append(("layer%d.%d" % (li + 1, bi), inpl, p, stride, stride != 1 or inpl != p))
            inpl = p
    return out


def _cp(c):
    return 4 if c < 4 else c


class SeResNet34Backend:
    graph_safe = True           # a train step makes no host decision and draws its masks from device-side counters: OcTrainer may replay it from a HIP graph

    def __init__(self, params=None, device="cuda", seed=1, compute="f32"):
        """compute: "f32" (exact-f32 MFMA, the parity path) or "bf16" (operands rounded to bf16 on the way into LDS, bf16 MFMA, f32
        accumulate; weight gradients likewise) -- activations, parameters and gradients stay f32 in memory either way."""
        require_gpu()
        if compute not in ("f32", "bf16"):
            raise OccError("compute must be 'f32' or 'bf16'")
        self.compute = compute
        self.device = torch.device(device)
        self.table = senet_param_table()
        self.slots, off = {}, 0
        for name, shp in self.table:
            if len(shp) == 4 and shp[2:] != (1, 1):
                ishp = (shp[0], shp[2], shp[3], _cp(shp[1]))
            elif len(shp) == 4:
                ishp = (shp[0], shp[1])
            else:
                ishp = tuple(shp)
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
        for name, shp in self.table:
            if name.endswith("bn1.weight") or name.endswith("bn2.weight") or name.endswith("downsample.1.weight"):
                pre = name[:-len(".weight")]
                self.buf[pre + ".running_mean"] = torch.zeros(shp[0], device=self.device)
                self.buf[pre + ".running_var"] = torch.ones(shp[0], device=self.device)
                self.buf[pre + ".num_batches_tracked"] = torch.zeros(1, device=self.device, dtype=torch.int64)
        self.bn_ws = torch.empty(512 * 256 * 2, device=self.device, dtype=torch.float64)
        self.bn_sums = torch.empty(512, device=self.device)
        self._ws, self.ctx = {}, None
        self.load_reference_params(params if params is not None else synthetic_senet_params(seed))

    # ---------------------------------------------------------------------------- checkpoint layouts --
    def load_reference_params(self, sd):
        with torch.no_grad():
            for name, shp in self.table:
                if name not in sd:
                    raise OccError("SE-ResNet state_dict lacks %s" % name)
                v = sd[name].detach().to(self.device, torch.float32)
                if tuple(v.shape) != tuple(shp):
                    raise OccError("shape of %s is %s, expected %s" % (name, tuple(v.shape), tuple(shp)))
                if len(shp) == 4 and shp[2:] != (1, 1):
                    self.p[name].zero_(); self.p[name][..., : shp[1]].copy_(v.permute(0, 2, 3, 1))
                else:
                    self.p[name].copy_(v.reshape(self.p[name].shape))
            for k in self.buf:
                if k in sd:
                    self.buf[k].copy_(sd[k].to(self.device).reshape(self.buf[k].shape))

    def ref_views(self, flat):
        """{reference name: VIEW of ``flat`` (self.P / self.G / same-sized buffer) in the reference's shape} (conv weights are
        channels-last and channel-padded inside)."""
        out = {}
        for name, shp in self.table:
            o, ishp, n = self.slots[name]
            v = flat[o:o + n].view(ishp)
            out[name] = v[..., : shp[1]].permute(0, 3, 1, 2) if len(shp) == 4 and shp[2:] != (1, 1) else v.view(shp)
        return out

    def _export(self, src):
        out = {}
        for name, shp in self.table:
            if len(shp) == 4 and shp[2:] != (1, 1):
                out[name] = src[name][..., : shp[1]].permute(0, 3, 1, 2).contiguous().clone()
            else:
                out[name] = src[name].detach().clone().reshape(shp)
        return out

    def state_dict(self):
        out = self._export(self.p)
        for k, v in self.buf.items():
            out[k] = v.clone().reshape(()) if k.endswith("num_batches_tracked") else v.clone()
        return out

    def grad_dict(self):
        return self._export(self.g)

    def zero_grad(self):
        K.fill(self.G, 0.0)

    # --------------------------------------------------------------------------------------- helpers --
    def _gemm(self, M, N, Kd, A, a_map, W, ldw, C, c_map, c_dtype, ab_dtype, a_seg=None, **kw):
        """occ_gemm; in bf16 compute mode the f32 operands are rounded to bf16 on their way into LDS (bf16 MFMA, f32 accumulate)
        wherever K and the operand strides are multiples of 8 elements -- everything but the 4-channel stem convolution."""
        if self.compute == "bf16" and ab_dtype == OCC_F32 and Kd % 8 == 0 and ldw % 8 == 0 and a_map.row_stride % 8 == 0 and \
                a_map.batch_stride % 8 == 0 and a_map.line_stride % 8 == 0 and (a_seg is None or (a_seg[1] % 8 == 0 and a_seg[2] % 8 == 0)):
            ab_dtype = OCC_F32_AS_BF16
        return ops.gemm_raw(M, N, Kd, A, a_map, W, ldw, C, c_map, c_dtype, ab_dtype, a_seg=a_seg, **kw)

    def _tn(self, *a, **kw):
        return K.gemm_tn(*a, bf16_mfma=(self.compute == "bf16"), **kw)

    def _e(self, *s, dtype=torch.float32):
        return torch.empty(*s, device=self.device, dtype=dtype)

    def _z(self, *s):
        return torch.zeros(*s, device=self.device)

    def _bn_fwd(self, name, x, rows, C, act, y, y_map, train, c):
        mean, rstd = self._e(C), self._e(C)
        K.bn_stats(x, rowmap(rows, 0, C), rows, C, self.bn_ws, mean, rstd, self.buf[name + ".running_mean"], self.buf[name + ".running_var"],
                   self.buf[name + ".num_batches_tracked"], train)
        K.bn_act_fwd(x, rowmap(rows, 0, C), mean, rstd, self.p[name + ".weight"], self.p[name + ".bias"], act, y, y_map, rows, C)
        c[name] = (mean, rstd)

    def _bn_bwd(self, name, dy, dy_map, x, rows, C, act, dx, dx_map, c):
        mean, rstd = c[name]
        K.bn_act_bwd(dy, dy_map, x, rowmap(rows, 0, C), mean, rstd, self.p[name + ".weight"], self.p[name + ".bias"], act, dx, dx_map,
                     self.g[name + ".weight"], self.g[name + ".bias"], self.bn_ws, self.bn_sums, rows, C)

    def _geometry(self, B, H0, W0):
        key = (B, H0, W0)
        if key in self._ws:
            return self._ws[key]
        g = {"H0": H0, "W0": W0}
        g["X0"] = self._z(B, H0 + 6, W0 + 6, 4)
        H1, W1 = (H0 + 6 - 7) // 2 + 1, (W0 + 6 - 7) // 2 + 1
        g["c1hw"] = (H1, W1)
        H, W = (H1 + 2 - 3) // 2 + 1, (W1 + 2 - 3) // 2 + 1
        g["blocks"] = []
        for pre, inp, p, s, ds in _blocks():
            Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
            bg = {"pre": pre, "inp": inp, "p": p, "s": s, "ds": ds, "H": H, "W": W, "Ho": Ho, "Wo": Wo,
                  "X": self._z(B, H + 2, W + 2, inp),             # block input with a 1-pixel zero border
                  "Y": self._z(B, Ho + 2, Wo + 2, p),             # relu(bn1(conv1)) with border
                  "D2": self._z(B, Ho + 2, Wo + 2, p),            # grad wrt conv2 output (border for the 3x3 dgrad)
                  "D1": self._z(B, Ho + 2, Wo + 2, p) if s == 1 else self._z(B, Ho + 1, Wo + 1, p)}   # grad wrt conv1 output
            g["blocks"].append(bg)
            H, W = Ho, Wo
        g["final_hw"] = (H, W)
        self._ws[key] = g
        return g

    # ======================================================================================= forward ==
    def forward(self, x, train=False, masks=None):
        """x f32 [B,1,T,D] (or [B,T,D] front-end features) -> (com [B,128], des [B,2]).  ``masks`` is accepted for interface parity with the
        AASIST engine and ignored: the network has no dropout."""
        x = x.to(self.device, torch.float32)
        if x.dim() == 3:
            x = x.unsqueeze(1)
        if x.dim() != 4 or x.shape[1] != 1:
            raise OccError("se_resnet34 expects [B,1,T,D]")
        B, _, H0, W0 = x.shape
        geo = self._geometry(B, H0, W0)
        c = {"B": B, "geo": geo, "train": train}
        p, es = self.p, 4
        X0 = geo["X0"]
        Wp0 = W0 + 6
        # place the image (channel 0 of 4) inside its 3-pixel border
        K.copy_rows(x.contiguous().view(-1), rowmap(B * H0 * W0, 0, 1), X0.data_ptr() + (3 * Wp0 + 3) * 4 * es, rowmap(H0 * W0, (H0 + 6) * Wp0 * 4, 4, W0, Wp0 * 4),
                    B * H0 * W0, 1)
        H1, W1 = geo["c1hw"]
        R0 = B * H1 * W1
        c1 = self._e(R0, 16)
        self._gemm(R0, 16, 7 * 7 * 4, X0, rowmap(H1 * W1, (H0 + 6) * Wp0 * 4, 2 * 4, W1, 2 * Wp0 * 4), p["conv1.weight"], 196, c1, rowmap(R0, 0, 16),
                     OCC_F32, OCC_F32, a_seg=(7, 28, Wp0 * 4))
        c["c1"] = c1
        a1 = self._e(R0, 16)
        self._bn_fwd("bn1", c1, R0, 16, ACT_RELU, a1, rowmap(R0, 0, 16), train, c)
        b0 = geo["blocks"][0]
        c["a1"] = a1
        c["pool_idx"] = torch.empty(B * b0["H"] * b0["W"] * 16, device=self.device, dtype=torch.uint8)
        Wp = b0["W"] + 2
        in_map = rowmap(b0["H"] * b0["W"], (b0["H"] + 2) * Wp * 16, 16, b0["W"], Wp * 16)
        check(lib().occ_maxpool3s2_fwd(ptr(a1), b0["X"].data_ptr() + (Wp + 1) * 16 * es, _rm(in_map), ptr(c["pool_idx"]), B, H1, W1, 16, stream_ptr()),
              "occ_maxpool3s2_fwd")
        nb = len(geo["blocks"])
        final = None
        for bi, bg in enumerate(geo["blocks"]):
            pre, inp, pl, s = bg["pre"], bg["inp"], bg["p"], bg["s"]
            H, W, Ho, Wo = bg["H"], bg["W"], bg["Ho"], bg["Wo"]
            Wp, Wop = W + 2, Wo + 2
            R = B * Ho * Wo
            X, Y = bg["X"], bg["Y"]
            o1 = self._e(R, pl)
            self._gemm(R, pl, 9 * inp, X, rowmap(Ho * Wo, (H + 2) * Wp * inp, s * inp, Wo, s * Wp * inp), p[pre + ".conv1.weight"], 9 * inp, o1, rowmap(R, 0, pl),
                         OCC_F32, OCC_F32, a_seg=(3, 3 * inp, Wp * inp))
            y_in = rowmap(Ho * Wo, (Ho + 2) * Wop * pl, pl, Wo, Wop * pl)
            self._bn_fwd(pre + ".bn1", o1, R, pl, ACT_RELU, Y.data_ptr() + (Wop + 1) * pl * es, y_in, train, c)
            o2 = self._e(R, pl)
            self._gemm(R, pl, 9 * pl, Y, rowmap(Ho * Wo, (Ho + 2) * Wop * pl, pl, Wo, Wop * pl), p[pre + ".conv2.weight"], 9 * pl, o2, rowmap(R, 0, pl), OCC_F32, OCC_F32,
                         a_seg=(3, 3 * pl, Wop * pl))
            y2 = self._e(R, pl)
            self._bn_fwd(pre + ".bn2", o2, R, pl, ACT_NONE, y2, rowmap(R, 0, pl), train, c)
            sq, z, gate = self._e(B, pl), self._e(B, pl // 16), self._e(B, pl)
            check(lib().occ_batch_colsum(ptr(y2), _rm(rowmap(B * Ho * Wo, 0, pl)), B, Ho * Wo, pl, 1.0 / (Ho * Wo), ptr(sq), stream_ptr()), "occ_batch_colsum")
            check(lib().occ_se_gate_fwd(ptr(sq), ptr(p[pre + ".se.fc.0.weight"]), ptr(p[pre + ".se.fc.2.weight"]), B, pl, pl // 16, ptr(z), ptr(gate), stream_ptr()),
                  "occ_se_gate_fwd")
            x_in = rowmap(H * W, (H + 2) * Wp * inp, inp, W, Wp * inp)
            if bg["ds"]:
                dso = self._e(R, pl)
                self._gemm(R, pl, inp, X.data_ptr() + (Wp + 1) * inp * es, rowmap(Ho * Wo, (H + 2) * Wp * inp, s * inp, Wo, s * Wp * inp),
                             p[pre + ".downsample.0.weight"], inp, dso, rowmap(R, 0, pl), OCC_F32, OCC_F32)
                res = self._e(R, pl)
                self._bn_fwd(pre + ".downsample.1", dso, R, pl, ACT_NONE, res, rowmap(R, 0, pl), train, c)
                res_addr, res_map = res, rowmap(R, 0, pl)
                c[pre + ".dso"] = dso
            else:
                res_addr, res_map = X.data_ptr() + (Wp + 1) * inp * es, x_in
            if bi + 1 < nb:
                nxt = geo["blocks"][bi + 1]
                nWp = nxt["W"] + 2
                out_addr = nxt["X"].data_ptr() + (nWp + 1) * pl * es
                out_map = rowmap(Ho * Wo, (Ho + 2) * nWp * pl, pl, Wo, nWp * pl)
            else:
                final = self._e(R, pl)
                out_addr, out_map = final, rowmap(R, 0, pl)
            check(lib().occ_se_scale_add_relu(ptr(y2), ptr(gate), K._a(res_addr), _rm(res_map), K._a(out_addr), _rm(out_map), B, Ho * Wo, pl, stream_ptr()),
                  "occ_se_scale_add_relu")
            c[pre] = dict(o1=o1, o2=o2, y2=y2, sq=sq, z=z, gate=gate, out_addr=out_addr, out_map=out_map, keep=(final,))
        Hf, Wf = geo["final_hw"]
        feat = self._e(B, 128)
        check(lib().occ_batch_colsum(ptr(final), _rm(rowmap(B * Hf * Wf, 0, 128)), B, Hf * Wf, 128, 1.0 / (Hf * Wf), ptr(feat), stream_ptr()), "occ_batch_colsum")
        com, des = self._e(B, 128), self._e(B, 4)
        self._gemm(B, 128, 128, feat, rowmap(B, 0, 128), p["embedding.weight"], 128, com, rowmap(B, 0, 128), OCC_F32, OCC_F32, bias=p["embedding.bias"])
        self._cls_w = self._cls_pad()
        self._gemm(B, 4, 128, feat, rowmap(B, 0, 128), self._cls_w[0], 128, des, rowmap(B, 0, 4), OCC_F32, OCC_F32, bias=self._cls_w[1])
        c["feat"], c["final"] = feat, final
        self.ctx = c if train else None
        return com, des[:, :2].contiguous()

    def _cls_pad(self):
        """classifier has 2 outputs; the GEMM wants N % 4 == 0 -> zero-padded copy [4,128] (+bias [4])."""
        w, b = self._z(4, 128), self._z(4)
        w[:2].copy_(self.p["classifier.weight"]); b[:2].copy_(self.p["classifier.bias"])
        return w, b

    # ====================================================================================== backward ==
    def _flip3(self, w, ci, co):
        """[co,3,3,ci] -> [ci,3,3,co] with both taps reversed (full-correlation operand of a stride-1 3x3 conv)."""
        out = self._e(ci, 3, 3, co)
        K.copy_strided(w, out, 2 * 3 * ci + 2 * ci, (ci, 3, 3, co), (1, -3 * ci, -ci, 9 * ci))
        return out

    def backward(self, dcom, ddes, want_dfeats=False):
        """Accumulates parameter gradients into G; with want_dfeats returns d loss / d input [B,T,D]."""
        c = self.ctx
        if c is None:
            raise OccError("backward() needs a preceding forward(train=True)")
        B, geo, p, g, es = c["B"], c["geo"], self.p, self.g, 4
        Hf, Wf = geo["final_hw"]
        feat = c["feat"]
        # ---- heads ---------------------------------------------------------------------------------------------------
        dd4 = self._z(B, 4); dd4[:, :2].copy_(ddes)
        gw4, gb4 = self._z(4, 128), self._z(4)
        self._tn(B, 128, 128, dcom.contiguous(), rowmap(B, 0, 128), feat, rowmap(B, 0, 128), g["embedding.weight"], 128, colsum_out=g["embedding.bias"])
        self._tn(B, 4, 128, dd4, rowmap(B, 0, 4), feat, rowmap(B, 0, 128), gw4, 128, colsum_out=gb4)
        K.axpby(gw4[:2].contiguous().view(-1), g["classifier.weight"].view(-1), g["classifier.weight"].view(-1))
        K.axpby(gb4[:2].contiguous(), g["classifier.bias"], g["classifier.bias"])
        wte = self._e(128, 128)
        K.copy_strided(p["embedding.weight"], wte, 0, (1, 1, 128, 128), (0, 0, 1, 128))
        wtc = self._e(128, 4)
        K.copy_strided(self._cls_w[0], wtc, 0, (1, 1, 128, 4), (0, 0, 1, 128))
        dfeat = self._e(B, 128)
        self._gemm(B, 128, 128, dcom.contiguous(), rowmap(B, 0, 128), wte, 128, dfeat, rowmap(B, 0, 128), OCC_F32, OCC_F32)
        self._gemm(B, 128, 4, dd4, rowmap(B, 0, 4), wtc, 4, dfeat, rowmap(B, 0, 128), OCC_F32, OCC_F32, R=dfeat, r_map=rowmap(B, 0, 128), r_dtype=OCC_F32)
        # grad wrt the last block's output: dfeat / (Hf*Wf) broadcast over the positions
        Rf = B * Hf * Wf
        dout = self._z(Rf, 128)
        K.axpby(dfeat.view(-1), None, dfeat.view(-1), 1.0 / (Hf * Wf), 0.0)
        check(lib().occ_add_batch_vec(ptr(dout), _rm(rowmap(Rf, 0, 128)), ptr(dfeat), B, Hf * Wf, 128, stream_ptr()), "occ_add_batch_vec")
        dout_addr, dout_map = dout, rowmap(Rf, 0, 128)
        blocks = geo["blocks"]
        for bi in range(len(blocks) - 1, -1, -1):
            bg = blocks[bi]
            pre, inp, pl, s = bg["pre"], bg["inp"], bg["p"], bg["s"]
            H, W, Ho, Wo = bg["H"], bg["W"], bg["Ho"], bg["Wo"]
            Wp, Wop = W + 2, Wo + 2
            R = B * Ho * Wo
            X, Y, D2, D1 = bg["X"], bg["Y"], bg["D2"], bg["D1"]
            sv = c[pre]
            # grad wrt this block's input, contiguous [B,H,W,inp]
            dX = self._e(B * H * W, inp)
            dy2, dgate = self._e(R, pl), self._z(B, pl)
            if bg["ds"]:
                dres, dres_map, dacc = self._e(R, pl), rowmap(R, 0, pl), 0
            else:
                dres, dres_map, dacc = dX, rowmap(B * H * W, 0, inp), 0          # identity shortcut: dX starts as dpre
            check(lib().occ_se_scale_add_relu_bwd(K._a(dout_addr), _rm(dout_map), K._a(sv["out_addr"]), _rm(sv["out_map"]), ptr(sv["y2"]), ptr(sv["gate"]), ptr(dy2),
                                                  ptr(dres), _rm(dres_map), dacc, ptr(dgate), B, Ho * Wo, pl, stream_ptr()), "occ_se_scale_add_relu_bwd")
            ds_ = self._e(B, pl)
            check(lib().occ_se_gate_bwd(ptr(sv["sq"]), ptr(sv["z"]), ptr(sv["gate"]), ptr(dgate), ptr(p[pre + ".se.fc.0.weight"]), ptr(p[pre + ".se.fc.2.weight"]),
                                        B, pl, pl // 16, ptr(g[pre + ".se.fc.0.weight"]), ptr(g[pre + ".se.fc.2.weight"]), ptr(ds_), stream_ptr()), "occ_se_gate_bwd")
            K.axpby(ds_.view(-1), None, ds_.view(-1), 1.0 / (Ho * Wo), 0.0)
            check(lib().occ_add_batch_vec(ptr(dy2), _rm(rowmap(R, 0, pl)), ptr(ds_), B, Ho * Wo, pl, stream_ptr()), "occ_add_batch_vec")
            # bn2 -> grad wrt conv2 output, into D2's interior
            d2_in = D2.data_ptr() + (Wop + 1) * pl * es
            y_in = rowmap(Ho * Wo, (Ho + 2) * Wop * pl, pl, Wo, Wop * pl)
            self._bn_bwd(pre + ".bn2", dy2, rowmap(R, 0, pl), sv["o2"], R, pl, ACT_NONE, d2_in, y_in, c)
            self._tn(R, pl, 9 * pl, d2_in, y_in, Y, rowmap(Ho * Wo, (Ho + 2) * Wop * pl, pl, Wo, Wop * pl), g[pre + ".conv2.weight"], 9 * pl, b_seg=(3, 3 * pl, Wop * pl))
            wd2 = self._flip3(p[pre + ".conv2.weight"], pl, pl)
            dY = self._e(R, pl)
            self._gemm(R, pl, 9 * pl, D2, rowmap(Ho * Wo, (Ho + 2) * Wop * pl, pl, Wo, Wop * pl), wd2, 9 * pl, dY, rowmap(R, 0, pl), OCC_F32, OCC_F32, a_seg=(3, 3 * pl, Wop * pl))
            # bn1 + relu -> grad wrt conv1 output, into D1
            if s == 1:
                d1_in, d1_map = D1.data_ptr() + (Wop + 1) * pl * es, y_in
            else:
                d1_in, d1_map = D1.data_ptr(), rowmap(Ho * Wo, (Ho + 1) * (Wo + 1) * pl, pl, Wo, (Wo + 1) * pl)
            self._bn_bwd(pre + ".bn1", dY, rowmap(R, 0, pl), sv["o1"], R, pl, ACT_RELU, d1_in, d1_map, c)
            self._tn(R, pl, 9 * inp, d1_in, d1_map, X, rowmap(Ho * Wo, (H + 2) * Wp * inp, s * inp, Wo, s * Wp * inp), g[pre + ".conv1.weight"], 9 * inp,
                      b_seg=(3, 3 * inp, Wp * inp))
            w1 = p[pre + ".conv1.weight"]                       # [pl,3,3,inp]
            full = rowmap(B * H * W, 0, inp)
            if s == 1:
                wd1 = self._flip3(w1, inp, pl)
                self._gemm(B * H * W, inp, 9 * pl, D1, rowmap(H * W, (Ho + 2) * Wop * pl, pl, W, Wop * pl), wd1, 9 * pl, dX, full, OCC_F32, OCC_F32,
                             a_seg=(3, 3 * pl, Wop * pl), R=None if bg["ds"] else dX, r_map=None if bg["ds"] else full, r_dtype=OCC_F32)
            else:
                self._dgrad_s2(bg, B, w1, D1, dX)
            if bg["ds"]:
                # shortcut: BN backward, weight gradient, and dX[2a][2c] += dres . Wds
                dso = c[pre + ".dso"]
                ddso = self._e(R, pl)
                self._bn_bwd(pre + ".downsample.1", dres, rowmap(R, 0, pl), dso, R, pl, ACT_NONE, ddso, rowmap(R, 0, pl), c)
                xs_map = rowmap(Ho * Wo, (H + 2) * Wp * inp, s * inp, Wo, s * Wp * inp)
                self._tn(R, pl, inp, ddso, rowmap(R, 0, pl), X.data_ptr() + (Wp + 1) * inp * es, xs_map, g[pre + ".downsample.0.weight"], inp)
                wdt = self._e(inp, pl)
                K.copy_strided(p[pre + ".downsample.0.weight"], wdt, 0, (1, 1, inp, pl), (0, 0, 1, inp))
                cmap = rowmap(Ho * Wo, H * W * inp, s * inp, Wo, s * W * inp)
                self._gemm(R, inp, pl, ddso, rowmap(R, 0, pl), wdt, pl, dX, cmap, OCC_F32, OCC_F32, R=dX, r_map=cmap, r_dtype=OCC_F32)
            dout_addr, dout_map = dX, full
        # ---- stem: max-pool, bn1+relu, 7x7 conv weight gradient ---------------------------------------------------------------
        H1, W1 = geo["c1hw"]
        R0 = B * H1 * W1
        da1 = self._z(R0, 16)
        b0 = blocks[0]
        check(lib().occ_maxpool3s2_bwd(ptr(dout_addr), _rm(rowmap(B * b0["H"] * b0["W"], 0, 16)), ptr(c["pool_idx"]), ptr(da1), B, H1, W1, 16, stream_ptr()),
              "occ_maxpool3s2_bwd")
        dc1 = self._e(R0, 16)
        self._bn_bwd("bn1", da1, rowmap(R0, 0, 16), c["c1"], R0, 16, ACT_RELU, dc1, rowmap(R0, 0, 16), c)
        H0, W0 = geo["H0"], geo["W0"]
        Wp0 = W0 + 6
        self._tn(R0, 16, 196, dc1, rowmap(R0, 0, 16), geo["X0"], rowmap(H1 * W1, (H0 + 6) * Wp0 * 4, 8, W1, 2 * Wp0 * 4), g["conv1.weight"], 196, b_seg=(7, 28, Wp0 * 4))
        self.ctx = None
        if want_dfeats:
            dx = self._e(B, H0, W0)
            check(lib().occ_conv7s2_dgrad_c1(ptr(dc1), ptr(p["conv1.weight"]), ptr(dx), B, H0, W0, stream_ptr()), "occ_conv7s2_dgrad_c1")
            return dx
        return None

    def _dgrad_s2(self, bg, B, w1, D1, dX):
        """Input gradient of a 3x3 stride-2 pad-1 conv as four parity-class GEMMs over D1 [B,Ho+1,Wo+1,pl] (zero last row / column)."""
        inp, pl, H, W, Ho, Wo = bg["inp"], bg["p"], bg["H"], bg["W"], bg["Ho"], bg["Wo"]
        ld = (Wo + 1) * pl
        bs = (Ho + 1) * ld
        Ha, Hb, Wa, Wb = (H + 1) // 2, H // 2, (W + 1) // 2, W // 2
        es = 4
        # w1 internal layout [co][kh][kw][ci]; operand rows = ci, columns = (window slot..., co)
        def pack(taps):              # taps: list of (kh, kw) in window-slot order
            out = self._e(inp, len(taps) * pl)
            for j, (kh, kw) in enumerate(taps):
                # out[ci][j*pl + co] = w1[co][kh][kw][ci]
                _copy_tap(w1, out, (kh * 3 + kw) * inp, inp, pl, 9 * inp, j * pl, len(taps) * pl)
            return out
        cls = [
            (0, 0, Ha, Wa, [(1, 1)], None),                                        # even row, even col: tap (1,1) at (a, c)
            (0, 1, Ha, Wb, [(1, 2), (1, 0)], ("cols", 2)),                         # even row, odd col: cols c (kw=2), c+1 (kw=0)
            (1, 0, Hb, Wa, [(2, 1), (0, 1)], ("rows", 2)),                         # odd row: rows a (kh=2), a+1 (kh=0)
            (1, 1, Hb, Wb, [(2, 2), (2, 0), (0, 2), (0, 0)], ("both", 2)),
        ]
        for eh, ew, Hc, Wc, taps, kind in cls:
            if Hc == 0 or Wc == 0:
                continue
            wop = pack(taps)
            Kd = len(taps) * pl
            a_map = rowmap(Hc * Wc, bs, pl, Wc, ld)
            seg = None
            if kind is not None and kind[0] in ("rows", "both"):
                seg = (2, Kd // 2, ld)
            c_map = rowmap(Hc * Wc, H * W * inp, 2 * inp, Wc, 2 * W * inp)
            self._gemm(B * Hc * Wc, inp, Kd, D1, a_map, wop, Kd, dX.data_ptr() + (eh * W + ew) * inp * es, c_map, OCC_F32, OCC_F32, a_seg=seg)


def _copy_tap(src, dst, src_off, n_rows, n_cols, src_col_stride, dst_col_off, dst_ld):
    """dst[r][dst_col_off + c] = src[src_off + c*src_col_stride + r]  (r < n_rows, c < n_cols): one tap of a conv weight, transposed."""
    sh = (1, 1, n_rows, n_cols)
    # copy_strided writes a CONTIGUOUS destination; go through a temporary and a row copy into the column block
    tmp = torch.empty(n_rows, n_cols, device=dst.device, dtype=torch.float32)
    K.copy_strided(src, tmp, src_off, sh, (0, 0, 1, src_col_stride))
    K.copy_rows(tmp, rowmap(n_rows, 0, n_cols), dst.data_ptr() + dst_col_off * 4, rowmap(n_rows, 0, dst_ld), n_rows, n_cols)



def synthetic_senet_params(seed=1):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shp in sorted(senet_param_table()):
        r = torch.randn(shp, generator=g)
        leaf = name.rsplit(".", 1)[-1]
        if len(shp) <= 1 and leaf == "weight":
            out[name] = 1.0 + 0.1 * r
        elif len(shp) <= 1:
            out[name] = 0.05 * r
        else:
            fan = 1
            for d in shp[1:]:
                fan *= d
            out[name] = r / math.sqrt(fan)
    return out


class _SeResNet(AliasGuard, torch.nn.Module):
    """Module facade with the reference's call signature (senet.py:120-142).  Every reference tensor is a registered nn.Parameter
    (reference name and shape, a view of the engine's flat buffer) and a training-mode forward is taped through
    ``autograd_bridge.EngineFunction``: ``optim.Adam(m.parameters())`` / ``loss.backward()`` / ``optimizer.step()`` of
    test_dataloader_v2.py:69, 119-130 drive the HIP kernels."""

    def __init__(self, state_dict=None, device="cuda", seed=1, compute="f32", **kwargs):
        super().__init__()
        self.backend = SeResNet34Backend(state_dict, device=device, seed=seed, compute=compute)
        self.param_set = attach_parameters(self, self.backend)

    def forward(self, x, eval=False):
        be = self.backend
        if not self.training:
            with torch.no_grad():
                return be.forward(x, train=False)

        def bwd(grads, needs):
            dcom, ddes = (None if g is None else g.contiguous().float() for g in grads)
            if dcom is None:
                dcom = torch.zeros(x.shape[0], 128, device=be.device)
            if ddes is None:
                ddes = torch.zeros(x.shape[0], 2, device=be.device)
            d = be.backward(dcom, ddes, want_dfeats=bool(needs[0]))
            return (d.view(x.shape) if needs[0] else None,)

        return run_engine(self.param_set, lambda t: be.forward(t, train=True), bwd, x)

    def backward(self, dcom, ddes, want_dfeats=False):
        return self.backend.backward(dcom, ddes, want_dfeats=want_dfeats)

    def state_dict(self, *a, **kw):
        return self.backend.state_dict()

    def load_state_dict(self, sd, strict=True):
        self.backend.load_reference_params(sd)
        return self

    def cuda(self, *a, **kw):
        return self

    def to(self, *a, **kw):
        return self


def se_resnet34(**kwargs):
    """senet.py:154-156."""
    return _SeResNet(**kwargs)


class ssl_resnet34(torch.nn.Module):
    """senet.py:162-185: XLS-R features [B,T,1024] -> unsqueeze(1) -> SE-ResNet34."""

    def __init__(self, device="cuda", ssl_cfg=None, ssl_dtype=torch.bfloat16, ssl_state_dict=None, state_dict=None, finetune_ssl=False, seed=1,
                 backend_compute=None, ssl_cp_path=None, synthetic_ssl=False):
        super().__init__()
        from .xlsr import SSLModel
        self.frontend = SSLModel(device, cp_path=ssl_cp_path, state_dict=ssl_state_dict, cfg=ssl_cfg, dtype=ssl_dtype, finetune=finetune_ssl,
                                 synthetic=synthetic_ssl)
        if backend_compute is None:         # same rule as AModel: a bf16 front-end brings the bf16-MFMA back-end mode
            backend_compute = "bf16" if ssl_dtype == torch.bfloat16 else "f32"
        self.resnet34 = se_resnet34(state_dict=state_dict, device=device, seed=seed, compute=backend_compute)
        # the names OcTrainer drives (shared with AModel)
        self.ssl_model = self.frontend
        self.backend = self.resnet34.backend

    def forward(self, x):
        feats = self.frontend.extract_feat(x)
        self.resnet34.train(self.training)
        return self.resnet34(feats.unsqueeze(1))

    def state_dict(self, *a, **kw):
        """Keys as the reference module tree gives them (senet.py:165-166): ``resnet34.*`` and ``frontend.model.*`` (fairseq names)."""
        sd = {"resnet34." + k: v for k, v in self.resnet34.state_dict().items()}
        for k, v in self.frontend.full_state_dict().items():
            sd["frontend.model." + k] = v.detach().clone() if torch.is_tensor(v) else v
        return sd

    def load_state_dict(self, sd, strict=True):
        self.resnet34.load_state_dict({k[len("resnet34."):]: v for k, v in sd.items() if k.startswith("resnet34.")})
        ssl = {k[len("frontend.model."):]: v for k, v in sd.items() if k.startswith("frontend.model.")}
        if ssl or strict:
            self.frontend.load_params(ssl, strict=strict)
        return self


class LfccFrontend:
    """Fixed (parameter-free) LFCC features: f32 waveform [B, L] -> f32 [B, n_frames, 13] (occm_amd.utils; reference utils.py:127-138)."""
    out_dim = 13

    def __init__(self, sr=16000):
        self.sr = sr

    def forward(self, wav, out_dtype=None, **kw):
        from ..utils import extract_lfcc_batch
        return extract_lfcc_batch(wav.contiguous(), self.sr)


class lfcc_resnet34(torch.nn.Module):
    """BASELINE configs[0]: LFCC [B,1,266,13] -> SE-ResNet34 -> (com [B,128], des [B,2]).  The reference has the two pieces
    (utils.extract_lfcc; models/senet.py se_resnet34, whose SURVEY shape probe is exactly [B,1,266,13]) but no entry point that joins
    them; this wrapper is that composition with the attribute names OcTrainer drives."""

    def __init__(self, device="cuda", state_dict=None, seed=1, sr=16000, backend_compute="f32"):
        super().__init__()
        import types
        self.frontend = LfccFrontend(sr)
        self.resnet34 = se_resnet34(state_dict=state_dict, device=device, seed=seed, compute=backend_compute)
        self.ssl_model = types.SimpleNamespace(model=self.frontend)
        self.backend = self.resnet34.backend

    def forward(self, x):
        self.resnet34.train(self.training)
        return self.resnet34(self.frontend.forward(x).unsqueeze(1))

    def state_dict(self, *a, **kw):
        return {"resnet34." + k: v for k, v in self.resnet34.state_dict().items()}

    def load_state_dict(self, sd, strict=True):
        self.resnet34.load_state_dict({k[len("resnet34."):]: v for k, v in sd.items() if k.startswith("resnet34.")})
        return self
