"""XLS-R / wav2vec 2.0 front-end on MI355X -- drop-in for ``models/xlsr.py`` (SSLModel :25-52) and the
``SSLModel`` of ``models/sslassist.py:20-49``.

Reference behaviour: ``extract_feat(x[B,L] or [B,L,1]) -> [B,T,out_dim]`` = fairseq
``Wav2Vec2Model(source, mask=False, features_only=True)['x']``.  Here the whole forward runs in HIP
kernels behind libocc_hip.so (occ_conv0_ln_gelu, occ_gemm, occ_layernorm, occ_attention); torch only
owns the buffers.  Parameters use fairseq's state_dict names so reference checkpoints
(``ssl_model.model.*`` keys of ``aasist_vocoded_{epoch}.pt``) load without renaming.

Activation layout in HBM: channels-last ``[B, T_i, C]`` throughout (a Conv1d window is then one
contiguous K-segment of the implicit GEMM); the residual stream is f32, GEMM operands are bf16 (or f32
for the parity path); the projection output is written into a zero-padded ``[B, T+128, D]`` buffer that
the grouped positional conv reads as 128 K-segments.
"""
import torch

from .. import ops
from .._lib import ACT_GELU, ACT_NONE, OCC_F32, OccError, dtype_code, require_gpu
from ..ops import rowmap

CONV_LAYERS = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2     # fairseq conv_feature_layers of XLS-R


class XlsrConfig:
    def __init__(self, dim=1024, ffn=4096, heads=16, layers=24, conv_dim=512, pos_k=128, pos_groups=16):
        self.dim, self.ffn, self.heads, self.layers = dim, ffn, heads, layers
        self.conv_dim, self.pos_k, self.pos_groups = conv_dim, pos_k, pos_groups

    @staticmethod
    def xlsr_300m():
        return XlsrConfig()

    @staticmethod
    def xlsr_1b():
        return XlsrConfig(dim=1280, ffn=5120, heads=16, layers=48)


def n_frames(L):
    for _, k, s in CONV_LAYERS:
        L = (L - k) // s + 1
    return L


def param_shapes(cfg):
    """fairseq Wav2Vec2Model state_dict names used by features_only forward."""
    s = {}
    cin = 1
    for i, (c, k, _) in enumerate(CONV_LAYERS):
        pre = "feature_extractor.conv_layers.%d" % i
        s[pre + ".0.weight"] = (c, cin, k); s[pre + ".0.bias"] = (c,)
        s[pre + ".2.1.weight"] = (c,); s[pre + ".2.1.bias"] = (c,)
        cin = c
    s["layer_norm.weight"] = (cfg.conv_dim,); s["layer_norm.bias"] = (cfg.conv_dim,)
    s["post_extract_proj.weight"] = (cfg.dim, cfg.conv_dim); s["post_extract_proj.bias"] = (cfg.dim,)
    s["encoder.pos_conv.0.weight_g"] = (1, 1, cfg.pos_k)
    s["encoder.pos_conv.0.weight_v"] = (cfg.dim, cfg.dim // cfg.pos_groups, cfg.pos_k)
    s["encoder.pos_conv.0.bias"] = (cfg.dim,)
    for i in range(cfg.layers):
        pre = "encoder.layers.%d" % i
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[pre + ".self_attn.%s.weight" % n] = (cfg.dim, cfg.dim)
            s[pre + ".self_attn.%s.bias" % n] = (cfg.dim,)
        s[pre + ".self_attn_layer_norm.weight"] = (cfg.dim,); s[pre + ".self_attn_layer_norm.bias"] = (cfg.dim,)
        s[pre + ".fc1.weight"] = (cfg.ffn, cfg.dim); s[pre + ".fc1.bias"] = (cfg.ffn,)
        s[pre + ".fc2.weight"] = (cfg.dim, cfg.ffn); s[pre + ".fc2.bias"] = (cfg.dim,)
        s[pre + ".final_layer_norm.weight"] = (cfg.dim,); s[pre + ".final_layer_norm.bias"] = (cfg.dim,)
    s["encoder.layer_norm.weight"] = (cfg.dim,); s["encoder.layer_norm.bias"] = (cfg.dim,)
    return s


class XlsrFrontend:
    """Frozen-weight forward engine.  ``params``: {fairseq name: tensor} (any device); ``dtype``:
    torch.bfloat16 (MFMA bf16, f32 accumulate) or torch.float32 (exact-f32 MFMA parity path)."""

    def __init__(self, params, cfg, device="cuda", dtype=torch.bfloat16):
        require_gpu()
        if dtype not in (torch.bfloat16, torch.float32):
            raise OccError("XlsrFrontend dtype must be bfloat16 or float32")
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        self.out_dim = cfg.dim
        self._ws = {}
        self.pack(params)

    # -- weight packing (host-side layout work, once per weight update) --------------------------
    def pack(self, p):
        dev, dt = self.device, self.dtype
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()
        cd = lambda t: t.detach().to(dev, torch.float32).to(dt).contiguous()
        w = {}
        w["c0.w"] = f32(p["feature_extractor.conv_layers.0.0.weight"]).reshape(512, -1).contiguous()
        for i, (c, k, s) in enumerate(CONV_LAYERS):
            pre = "feature_extractor.conv_layers.%d" % i
            w["c%d.b" % i] = f32(p[pre + ".0.bias"])
            w["c%d.g" % i] = f32(p[pre + ".2.1.weight"]); w["c%d.be" % i] = f32(p[pre + ".2.1.bias"])
            if i > 0:   # [Cout, Cin, k] -> [Cout, k*Cin]: K index = tap*Cin + channel (channels-last window)
                w["c%d.w" % i] = cd(p[pre + ".0.weight"].permute(0, 2, 1).reshape(c, -1))
        w["ln.g"] = f32(p["layer_norm.weight"]); w["ln.b"] = f32(p["layer_norm.bias"])
        w["proj.w"] = cd(p["post_extract_proj.weight"]); w["proj.b"] = f32(p["post_extract_proj.bias"])
        v = p["encoder.pos_conv.0.weight_v"].detach().to(torch.float32)
        g = p["encoder.pos_conv.0.weight_g"].detach().to(torch.float32)
        wn = g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()            # weight_norm(dim=2)
        G = self.cfg.pos_groups
        cg = self.cfg.dim // G
        # [D, cg, k] -> per group [cg_out, k*cg_in]: K index = tap*cg + channel
        w["pos.w"] = cd(wn.reshape(G, cg, cg, self.cfg.pos_k).permute(0, 1, 3, 2).reshape(G, cg, self.cfg.pos_k * cg))
        w["pos.b"] = f32(p["encoder.pos_conv.0.bias"])
        for i in range(self.cfg.layers):
            pre = "encoder.layers.%d" % i
            a = pre + ".self_attn."
            w["l%d.qkv.w" % i] = cd(torch.cat([p[a + "q_proj.weight"], p[a + "k_proj.weight"], p[a + "v_proj.weight"]], 0))
            w["l%d.qkv.b" % i] = f32(torch.cat([p[a + "q_proj.bias"], p[a + "k_proj.bias"], p[a + "v_proj.bias"]], 0))
            w["l%d.o.w" % i] = cd(p[a + "out_proj.weight"]); w["l%d.o.b" % i] = f32(p[a + "out_proj.bias"])
            w["l%d.ln1.g" % i] = f32(p[pre + ".self_attn_layer_norm.weight"]); w["l%d.ln1.b" % i] = f32(p[pre + ".self_attn_layer_norm.bias"])
            w["l%d.fc1.w" % i] = cd(p[pre + ".fc1.weight"]); w["l%d.fc1.b" % i] = f32(p[pre + ".fc1.bias"])
            w["l%d.fc2.w" % i] = cd(p[pre + ".fc2.weight"]); w["l%d.fc2.b" % i] = f32(p[pre + ".fc2.bias"])
            w["l%d.ln2.g" % i] = f32(p[pre + ".final_layer_norm.weight"]); w["l%d.ln2.b" % i] = f32(p[pre + ".final_layer_norm.bias"])
        w["enc_ln.g"] = f32(p["encoder.layer_norm.weight"]); w["enc_ln.b"] = f32(p["encoder.layer_norm.bias"])
        self.w = w

    # -- activation workspace, cached per (B, L) -------------------------------------------------
    def _workspace(self, B, L, slot=0):
        key = (B, L, slot)
        if key not in self._ws:
            dev, dt, cfg = self.device, self.dtype, self.cfg
            Ts, Lc = [], L
            for _, k, s in CONV_LAYERS:
                Lc = (Lc - k) // s + 1
                Ts.append(Lc)
            T = Ts[-1]
            if T < 1:
                raise OccError("waveform of %d samples is too short for the conv stack" % L)
            M = B * T
            ws = {"Ts": Ts, "T": T, "M": M}
            ws["cA"] = torch.empty(B * Ts[0] * 512, device=dev, dtype=dt)
            ws["cB"] = torch.empty(B * Ts[1] * 512, device=dev, dtype=dt)
            ws["feat"] = torch.empty(M, 512, device=dev, dtype=dt)
            ws["xpad"] = torch.zeros(B, T + cfg.pos_k, cfg.dim, device=dev, dtype=dt)
            ws["x"] = torch.empty(M, cfg.dim, device=dev, dtype=torch.float32)      # residual stream
            ws["h"] = torch.empty(M, cfg.dim, device=dev, dtype=dt)
            ws["qkv"] = torch.empty(M, 3 * cfg.dim, device=dev, dtype=dt)
            ws["att"] = torch.empty(M, cfg.dim, device=dev, dtype=dt)
            ws["ffn"] = torch.empty(M, cfg.ffn, device=dev, dtype=dt)
            self._ws[key] = ws
        return self._ws[key]

    def forward(self, wav, out_dtype=None, taps=None, slot=0, out=None):
        """wav f32 [B,L] on the GPU -> [B,T,dim] (dtype = out_dtype or the compute dtype).  ``slot`` selects an independent
        activation workspace so two half-batches can run concurrently on two streams; ``out`` is an optional destination."""
        cfg, w, dt = self.cfg, self.w, self.dtype
        dst = out
        if wav.dim() == 3:
            wav = wav[:, :, 0]                                   # sslassist.py:42-43
        wav = wav.to(self.device, torch.float32).contiguous()
        B, L = wav.shape
        ws = self._workspace(B, L, slot)
        Ts, T, M = ws["Ts"], ws["T"], ws["M"]
        code = dtype_code(ws["h"])
        D = cfg.dim
        # conv block 0 fused with its LayerNorm + GELU
        cur, nxt = ws["cA"], ws["cB"]
        ops.conv0_ln_gelu(wav, w["c0.w"], w["c0.b"], w["c0.g"], w["c0.be"], 10, 5, dt, out=cur[: B * Ts[0] * 512].view(B, Ts[0], 512))
        Tin = Ts[0]
        for i in range(1, 7):
            _, k, s = CONV_LAYERS[i]
            Tout = Ts[i]
            out = nxt[: B * Tout * 512].view(B * Tout, 512)
            ops.gemm_raw(B * Tout, 512, k * 512, cur, rowmap(Tout, Tin * 512, s * 512), w["c%d.w" % i], k * 512,
                         out, rowmap(B * Tout, 0, 512), code, code, bias=w["c%d.b" % i])
            ops.layernorm(out, w["c%d.g" % i], w["c%d.be" % i], gelu=True, out=out)
            cur, nxt = nxt, cur
            Tin = Tout
        feat = cur[: M * 512].view(M, 512)
        if taps is not None:
            taps["conv"] = feat.view(B, T, 512).float().clone()
        ops.layernorm(feat, w["ln.g"], w["ln.b"], out=ws["feat"])
        # post_extract_proj -> rows [pos_k/2, pos_k/2 + T) of the zero-padded buffer
        xpad = ws["xpad"]
        Tp = T + cfg.pos_k
        half = cfg.pos_k // 2
        es = xpad.element_size()
        inner = xpad.data_ptr() + half * D * es
        pmap = rowmap(T, Tp * D, D)
        ops.gemm_raw(M, D, 512, ws["feat"], rowmap(M, 0, 512), w["proj.w"], 512, inner, pmap, code, code, bias=w["proj.b"])
        # grouped positional conv (+bias, GELU) added to the projection -> f32 residual stream
        G = cfg.pos_groups
        cg = D // G
        x = ws["x"]
        ops.gemm_raw(M, cg, cfg.pos_k * cg, xpad, pmap, w["pos.w"], cfg.pos_k * cg, x, rowmap(M, 0, D), OCC_F32, code,
                     bias=w["pos.b"], act=ACT_GELU, R=inner, r_map=pmap, r_dtype=code,
                     a_seg=(cfg.pos_k, cg, D), groups=(G, cg, cg * cfg.pos_k * cg, cg))
        if taps is not None:
            taps["pos"] = x.view(B, T, D).clone()
        hd = D // cfg.heads
        xmap = rowmap(M, 0, D)
        for i in range(cfg.layers):
            ops.layernorm(x, w["l%d.ln1.g" % i], w["l%d.ln1.b" % i], out=ws["h"])
            ops.linear(ws["h"], w["l%d.qkv.w" % i], w["l%d.qkv.b" % i], out=ws["qkv"])
            ops.attention(ws["qkv"], B, T, cfg.heads, hd, hd ** -0.5, out=ws["att"])
            ops.gemm_raw(M, D, D, ws["att"], xmap, w["l%d.o.w" % i], D, x, xmap, OCC_F32, code, bias=w["l%d.o.b" % i],
                         R=x, r_map=xmap, r_dtype=OCC_F32)
            ops.layernorm(x, w["l%d.ln2.g" % i], w["l%d.ln2.b" % i], out=ws["h"])
            ops.linear(ws["h"], w["l%d.fc1.w" % i], w["l%d.fc1.b" % i], act=ACT_GELU, out=ws["ffn"])
            ops.gemm_raw(M, D, cfg.ffn, ws["ffn"], rowmap(M, 0, cfg.ffn), w["l%d.fc2.w" % i], cfg.ffn, x, xmap, OCC_F32, code,
                         bias=w["l%d.fc2.b" % i], R=x, r_map=xmap, r_dtype=OCC_F32)
            if taps is not None:
                taps["layer%d" % i] = x.view(B, T, D).clone()
        if dst is None:
            dst = torch.empty(B, T, D, device=self.device, dtype=out_dtype or dt)
        ops.layernorm(x, w["enc_ln.g"], w["enc_ln.b"], out=dst.view(M, D))
        return dst


class SSLModel(torch.nn.Module):
    """Mirror of models/xlsr.py:25-52 (and sslassist.py:20-49): ``SSLModel(device)``, ``.out_dim``,
    ``.extract_feat(x)``.  The reference hard-codes a fairseq checkpoint path (xlsr.py:29); here the
    weights come from ``state_dict`` (fairseq names), a ``cp_path`` torch file holding them, or -- when
    neither is given -- the deterministic synthetic filler (no checkpoint can exist offline)."""

    def __init__(self, device="cuda", cp_path=None, state_dict=None, cfg=None, dtype=torch.bfloat16, seed=0):
        super().__init__()
        self.device = device
        self.cfg = cfg or XlsrConfig.xlsr_300m()
        self.out_dim = self.cfg.dim
        if state_dict is None and cp_path is not None:
            ck = torch.load(cp_path, map_location="cpu")
            state_dict = ck.get("model", ck)
        if state_dict is None:
            state_dict = synthetic_params(self.cfg, seed)
        self._params = {k: v for k, v in state_dict.items() if k in param_shapes(self.cfg)}
        missing = set(param_shapes(self.cfg)) - set(self._params)
        if missing:
            raise OccError("XLS-R state_dict lacks %d tensors, e.g. %s" % (len(missing), sorted(missing)[:3]))
        self.model = XlsrFrontend(self._params, self.cfg, device=device, dtype=dtype)

    def extract_feat(self, input_data):
        with torch.no_grad():
            return self.model.forward(input_data)

    def forward(self, input_data):
        return self.extract_feat(input_data)


def synthetic_params(cfg, seed=0):
    """Deterministic stand-in weights (same generator family as oracle/fill.py, restated here because
    the product never imports oracle/)."""
    import math
    g = torch.Generator().manual_seed(seed)
    out = {}
    shapes = param_shapes(cfg)
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        leaf = name.rsplit(".", 1)[-1]
        r = torch.randn(shp, generator=g, dtype=torch.float32)
        if leaf == "weight_g":
            t = 1.0 + 0.1 * r
        elif len(shp) <= 1 and leaf == "weight":
            t = 1.0 + 0.1 * r
        elif len(shp) <= 1:
            t = 0.05 * r
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = r / math.sqrt(max(fan_in, 1))
        out[name] = t
    return out
