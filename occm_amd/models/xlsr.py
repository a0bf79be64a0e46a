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
import os

import torch

from .. import ops
from .._lib import ACT_GELU, ACT_GELU_GRAD, ACT_GELU_KEEP_GRAD, ACT_MUL_AUX, ACT_NONE, OCC_AF32_WBF16, OCC_BF16 as OCC_BF16_CODE, OCC_F32, OCC_F32X3, OccError, dtype_code, require_gpu
from ..autograd_bridge import attach_parameters, run_engine
from ..ops import rowmap

WS_CACHE = 12            # activation workspaces kept per model (one per distinct input shape)
# fc1's forward epilogue leaves bf16 gelu'(pre-activation) for backward (one exponential serves gelu and gelu') and fc2's input-gradient
# epilogue multiplies by it, instead of saving the pre-activation and evaluating gelu' there.  OCC_GELU_KEEP_GRAD=0: the older pair.
KEEP_GELU_GRAD = os.environ.get("OCC_GELU_KEEP_GRAD", "1") != "0"

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


class XlsrFrontend(torch.nn.Module):
    """Frozen-weight forward engine.  ``params``: {fairseq name: tensor} (any device); ``dtype``:
    torch.bfloat16 (MFMA bf16, f32 accumulate) or torch.float32 (exact-f32 MFMA parity path).

    An ``nn.Module`` only so that the fine-tuning subclasses can register their tensors as ``nn.Parameter``s under fairseq's names
    (``ssl_model.model.encoder.layers.0.fc1.weight`` ...), as the reference's module tree has them (sslassist.py:25-29); the frozen
    engine registers none (its operands are packed bf16 copies)."""

    def __init__(self, params, cfg, device="cuda", dtype=torch.bfloat16, f32_gemm="exact"):
        """f32_gemm (dtype float32 only): "exact" = the exact-f32 MFMA (v_mfma_f32_16x16x4_f32, the parity path); "x3" = every operand
        split into bf16 hi + lo while it is staged and three bf16 MFMAs per block (OCC_F32X3): products exact to 2^-16 relative at 3/16 of
        the cycles -- activations, LayerNorm, attention and accumulation stay f32 either way."""
        super().__init__()
        require_gpu()
        if dtype not in (torch.bfloat16, torch.float32):
            raise OccError("XlsrFrontend dtype must be bfloat16 or float32")
        if f32_gemm not in ("exact", "x3"):
            raise OccError("f32_gemm must be 'exact' or 'x3'")
        self.f32_gemm = f32_gemm
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
        if dt == torch.float32 and self.f32_gemm == "x3":
            # the transformer linears (82 % of the FLOPs) as ONE bf16 GEMM of depth 3K on the LDS-DMA kernels: weights [wh | wh | wl] here,
            # activations [xh | xl | xh] in front of every GEMM (forward)
            for i in range(self.cfg.layers):
                for n in ("qkv", "o", "fc1", "fc2"):
                    w["l%d.%s.w3" % (i, n)] = ops.split3_bf16(w["l%d.%s.w" % (i, n)], mode=1)
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
            if dt == torch.float32 and self.f32_gemm == "x3":
                ws["a3"] = torch.empty(M, 3 * max(cfg.dim, cfg.ffn), device=dev, dtype=torch.bfloat16)       # [hi | lo | hi] copy of a GEMM's input
            self._ws[key] = ws
            while len(self._ws) > WS_CACHE:            # variable-length scoring meets thousands of (B, L): keep the most recent shapes only
                self._ws.pop(next(iter(self._ws)))
        else:
            self._ws[key] = self._ws.pop(key)          # most recently used last
        return self._ws[key]

    def forward(self, wav, out_dtype=None, taps=None, slot=0, out=None, lengths=None):
        """wav f32 [B,L] on the GPU -> [B,T,dim] (dtype = out_dtype or the compute dtype).  ``slot`` selects an independent
        activation workspace so two half-batches can run concurrently on two streams; ``out`` is an optional destination.

        lengths (optional, samples per utterance of a ZERO-PADDED batch; scoring in length-sorted batches instead of the reference's
        one-utterance loop, oc_classifier.py:182-186): rows [0, n_frames(lengths[b])) of utterance b then equal its un-padded
        single-utterance result -- the un-padded conv stack never reads past a frame's own 400 samples, the projected features of the
        pad frames are zeroed before the positional conv (which sees zeros there in the un-padded run too) and attention masks the pad
        keys (occ_attention_varlen); every other layer is row-wise.  Rows past an utterance's frame count are not meaningful."""
        cfg, w, dt = self.cfg, self.w, self.dtype
        dst = out
        if wav.dim() == 3:
            wav = wav[:, :, 0]                                   # sslassist.py:42-43
        wav = wav.to(self.device, torch.float32).contiguous()
        B, L = wav.shape
        ws = self._workspace(B, L, slot)
        Ts, T, M = ws["Ts"], ws["T"], ws["M"]
        kv_len = None
        if lengths is not None:
            fr = [n_frames(int(v)) for v in lengths]
            if len(fr) != B or min(fr) < 1 or max(fr) > T:
                raise OccError("lengths must give, per utterance, a sample count between one frame's worth and the padded length")
            if min(fr) < T:
                kv_len = torch.tensor(fr, dtype=torch.int32, device=self.device)
        code = dtype_code(ws["h"])
        ab = OCC_F32X3 if (code == OCC_F32 and self.f32_gemm == "x3") else code          # the GEMMs' arithmetic (storage stays `code`)
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
                         out, rowmap(B * Tout, 0, 512), code, ab, bias=w["c%d.b" % i])
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
        ops.gemm_raw(M, D, 512, ws["feat"], rowmap(M, 0, 512), w["proj.w"], 512, inner, pmap, code, ab, bias=w["proj.b"])
        if kv_len is not None:                                   # the positional conv pads with zeros past the utterance's last frame
            xv = xpad.view(B, Tp, D)
            for b, tb in enumerate(fr):
                if tb < T:
                    xv[b, half + tb: half + T].zero_()
        # grouped positional conv (+bias, GELU) added to the projection -> f32 residual stream
        G = cfg.pos_groups
        cg = D // G
        x = ws["x"]
        ops.gemm_raw(M, cg, cfg.pos_k * cg, xpad, pmap, w["pos.w"], cfg.pos_k * cg, x, rowmap(M, 0, D), OCC_F32, ab,
                     bias=w["pos.b"], act=ACT_GELU, R=inner, r_map=pmap, r_dtype=code,
                     a_seg=(cfg.pos_k, cg, D), groups=(G, cg, cg * cfg.pos_k * cg, cg))
        if taps is not None:
            taps["pos"] = x.view(B, T, D).clone()
        hd = D // cfg.heads
        xmap = rowmap(M, 0, D)
        for i in range(cfg.layers):
            ops.layernorm(x, w["l%d.ln1.g" % i], w["l%d.ln1.b" % i], out=ws["h"])
            if "a3" in ws:
                self._layer_x3(i, ws, x, xmap, M, D, B, T, hd, kv_len)
                if taps is not None:
                    taps["layer%d" % i] = x.view(B, T, D).clone()
                continue
            ops.linear(ws["h"], w["l%d.qkv.w" % i], w["l%d.qkv.b" % i], out=ws["qkv"], ab_dtype=ab)
            if kv_len is not None:
                ops.attention_varlen(ws["qkv"], B, T, cfg.heads, hd, hd ** -0.5, kv_len, out=ws["att"])
            else:
                ops.attention(ws["qkv"], B, T, cfg.heads, hd, hd ** -0.5, out=ws["att"])
            ops.gemm_raw(M, D, D, ws["att"], xmap, w["l%d.o.w" % i], D, x, xmap, OCC_F32, ab, bias=w["l%d.o.b" % i],
                         R=x, r_map=xmap, r_dtype=OCC_F32)
            ops.layernorm(x, w["l%d.ln2.g" % i], w["l%d.ln2.b" % i], out=ws["h"])
            ops.linear(ws["h"], w["l%d.fc1.w" % i], w["l%d.fc1.b" % i], act=ACT_GELU, out=ws["ffn"], ab_dtype=ab)
            ops.gemm_raw(M, D, cfg.ffn, ws["ffn"], rowmap(M, 0, cfg.ffn), w["l%d.fc2.w" % i], cfg.ffn, x, xmap, OCC_F32, ab,
                         bias=w["l%d.fc2.b" % i], R=x, r_map=xmap, r_dtype=OCC_F32)
            if taps is not None:
                taps["layer%d" % i] = x.view(B, T, D).clone()
        if dst is None:
            dst = torch.empty(B, T, D, device=self.device, dtype=out_dtype or dt)
        ops.layernorm(x, w["enc_ln.g"], w["enc_ln.b"], out=dst.view(M, D))
        return dst


    def _x3(self, ws, a, M, K, wname, N, C, c_map, **kw):
        """C = act(a . W^T + bias) (+ R) with f32-grade products at the bf16 kernels' rate: a (f32 [M, K]) -> [ah | al | ah], W stored as
        [wh | wh | wl]: ONE bf16 GEMM of depth 3K = ah.wh + al.wh + ah.wl, f32 accumulate, f32 result."""
        a3 = ws["a3"].view(-1)[: M * 3 * K].view(M, 3 * K)
        ops.split3_bf16(a, out=a3, mode=0, rows=M, K=K)
        ops.gemm_raw(M, N, 3 * K, a3, rowmap(M, 0, 3 * K), self.w[wname], 3 * K, C, c_map, OCC_F32, OCC_BF16_CODE, **kw)

    def _layer_x3(self, i, ws, x, xmap, M, D, B, T, hd, kv_len):
        """One pre-LN transformer layer of the f32 path with split-operand linears (ws["h"] already holds LN1(x))."""
        cfg, w = self.cfg, self.w
        self._x3(ws, ws["h"], M, D, "l%d.qkv.w3" % i, 3 * D, ws["qkv"], rowmap(M, 0, 3 * D), bias=w["l%d.qkv.b" % i])
        if kv_len is not None:
            ops.attention_varlen(ws["qkv"], B, T, cfg.heads, hd, hd ** -0.5, kv_len, out=ws["att"])
        else:
            ops.attention(ws["qkv"], B, T, cfg.heads, hd, hd ** -0.5, out=ws["att"])
        self._x3(ws, ws["att"], M, D, "l%d.o.w3" % i, D, x, xmap, bias=w["l%d.o.b" % i], R=x, r_map=xmap, r_dtype=OCC_F32)
        ops.layernorm(x, w["l%d.ln2.g" % i], w["l%d.ln2.b" % i], out=ws["h"])
        self._x3(ws, ws["h"], M, D, "l%d.fc1.w3" % i, cfg.ffn, ws["ffn"], rowmap(M, 0, cfg.ffn), bias=w["l%d.fc1.b" % i], act=ACT_GELU)
        self._x3(ws, ws["ffn"], M, cfg.ffn, "l%d.fc2.w3" % i, D, x, xmap, bias=w["l%d.fc2.b" % i], R=x, r_map=xmap, r_dtype=OCC_F32)


class SSLModel(torch.nn.Module):
    """Mirror of models/xlsr.py:25-52 (and sslassist.py:20-49): ``SSLModel(device)``, ``.out_dim``,
    ``.extract_feat(x)``.  The reference hard-codes a fairseq checkpoint path (xlsr.py:29) and fails when the file is absent;
    here the weights come from ``state_dict`` (fairseq names) or a ``cp_path`` torch file holding them (a fairseq-shaped
    ``{"model": ..., "cfg": ...}`` file or a bare state dict) and a missing checkpoint is an error too.  ``synthetic=True`` is the
    explicit opt-in (tests, bench.py: no checkpoint can exist offline) to the deterministic filler.

    Tensors of the checkpoint that the features_only forward never touches (``mask_emb``, ``quantizer.*``, ``project_q.*``,
    ``final_proj.*`` ...) are kept in ``extra_state`` and re-emitted by ``full_state_dict`` so a file saved here still loads in the
    reference's strict ``load_state_dict`` (oc_classifier.py:340)."""

    _eval_at_init = True

    def __init__(self, device="cuda", cp_path=None, state_dict=None, cfg=None, dtype=torch.bfloat16, seed=0, finetune=False, synthetic=False,
                 train_cfg=None, f32_gemm="exact"):
        super().__init__()
        self.device = device
        self.cfg = cfg or XlsrConfig.xlsr_300m()
        self.out_dim = self.cfg.dim
        self.ckpt_cfg = None
        if state_dict is None and cp_path is not None:
            ck = torch.load(cp_path, map_location="cpu")
            state_dict = ck.get("model", ck)
            self.ckpt_cfg = ck.get("cfg") if isinstance(ck, dict) else None
        if state_dict is None:
            if not synthetic:
                raise OccError("SSLModel needs XLS-R weights: pass cp_path= / state_dict= (fairseq names), or synthetic=True for the "
                               "deterministic random filler (tests and bench.py only)")
            state_dict = synthetic_params(self.cfg, seed)
        shapes = param_shapes(self.cfg)
        self._params = {k: v for k, v in state_dict.items() if k in shapes}
        self.extra_state = {k: v for k, v in state_dict.items() if k not in shapes}
        check_param_shapes(self._params, shapes)
        self.finetune = finetune
        self.train_cfg = train_cfg or XlsrTrainCfg.from_checkpoint_cfg(self.ckpt_cfg)
        if finetune == "encoder":      # transformer encoder trainable, conv feature extractor frozen (feature_grad_mult = 0 style)
            self.model = XlsrFineTuner(self._params, self.cfg, device=device)
        elif finetune:                 # the reference's optimizer holds every SSL parameter (oc_training.py:324): end-to-end
            self.model = XlsrFullFineTuner(self._params, self.cfg, device=device)
        else:
            self.model = XlsrFrontend(self._params, self.cfg, device=device, dtype=dtype, f32_gemm=f32_gemm)
        if self._eval_at_init:
            self.model.eval()                  # xlsr.py:34 (sslassist.SSLModel has no such call: quirk 10 of SURVEY.md section 8a)
        self.param_set = None
        if finetune:
            self.model.train_cfg = self.train_cfg
            # every trainable tensor as an nn.Parameter named as in fairseq's module tree (a view of the flat master buffer), so that
            # ``optim.Adam(aasist.parameters())`` (oc_training.py:324) holds XLS-R too
            self.param_set = attach_parameters(self.model, self.model)

    def extract_feat(self, input_data, lengths=None):
        """sslassist.py:31-49 / xlsr.py:39-48.  ``lengths`` (not in the reference): sample counts of a zero-padded batch, see
        ``XlsrFrontend.forward`` -- deterministic forward only (frozen engine, or a fine-tuning engine in eval mode).  Frozen engine: a plain forward.  Fine-tuning engine in training mode: fairseq's
        train-mode forward (dropouts / layerdrop per ``train_cfg``), taped for ``loss.backward()`` through
        ``autograd_bridge.EngineFunction``; in eval mode the deterministic forward."""
        x = input_data[:, :, 0] if input_data.dim() == 3 else input_data
        eng = self.model
        if not self.finetune:
            with torch.no_grad():
                return eng.forward(x, out_dtype=torch.float32, lengths=lengths)
        eng.sync_operands()
        if not eng.training:                   # fairseq decides on the inner model's flag (``ssl.train()`` / ``aasist.train()`` set it)
            with torch.no_grad():
                return eng.forward(x, out_dtype=torch.float32, lengths=lengths)
        if lengths is not None:
            raise OccError("lengths (masked batches) are an evaluation feature: the reference trains on zero-padded groups un-masked (oc_training.py:244-249)")

        def bwd(grads, needs):
            eng.backward(grads[0].contiguous().float())
            return (None,)

        return run_engine(self.param_set, lambda w: eng.forward_train(w.to(eng.device, torch.float32).contiguous()), bwd, x)

    def forward(self, input_data, lengths=None):
        return self.extract_feat(input_data, lengths=lengths)

    def full_state_dict(self):
        """Every tensor of the loaded checkpoint under its fairseq name: the path's tensors (the trained values when fine-tuning) plus
        the untouched off-path ones."""
        out = dict(self._params)
        if self.finetune:
            out.update(self.model.export_params())
        out.update(self.extra_state)
        return out

    def load_params(self, sd, strict=True):
        """sd: {fairseq name: tensor}.  strict: every tensor of the path must be present with its shape; others are kept as extra state."""
        shapes = param_shapes(self.cfg)
        new = {k: v for k, v in sd.items() if k in shapes}
        if strict:
            check_param_shapes(new, shapes)
        else:
            check_param_shapes(new, {k: shapes[k] for k in new})
        self._params.update(new)
        self.extra_state.update({k: v for k, v in sd.items() if k not in shapes})
        if self.finetune:
            self.model._load_master(self._params)
            self.model.refresh_operands()
        else:
            self.model.pack(self._params)


def check_param_shapes(params, shapes):
    missing = sorted(set(shapes) - set(params))
    if missing:
        raise OccError("XLS-R state_dict lacks %d tensors, e.g. %s" % (len(missing), missing[:3]))
    bad = [(k, tuple(params[k].shape), tuple(shapes[k])) for k in shapes if tuple(params[k].shape) != tuple(shapes[k])]
    if bad:
        raise OccError("XLS-R state_dict has %d tensors of the wrong shape, e.g. %s is %s, expected %s" % ((len(bad),) + bad[0]))


class XlsrTrainCfg:
    """Train-mode behaviour of fairseq's Wav2Vec2Model that the reference switches on with ``aasist.train()`` (oc_training.py:351;
    sslassist.SSLModel never calls ``.eval()``, sslassist.py:20-29).  Values come from the checkpoint's ``cfg`` (fairseq: ``cfg.model``);
    the defaults are those of the published XLS-R pre-training configuration (all zero, feature_grad_mult 1.0)."""
    FIELDS = ("dropout", "attention_dropout", "activation_dropout", "encoder_layerdrop", "dropout_input", "dropout_features", "feature_grad_mult")

    def __init__(self, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, encoder_layerdrop=0.0, dropout_input=0.0, dropout_features=0.0,
                 feature_grad_mult=1.0):
        self.dropout, self.attention_dropout, self.activation_dropout = float(dropout), float(attention_dropout), float(activation_dropout)
        self.encoder_layerdrop, self.dropout_input, self.dropout_features = float(encoder_layerdrop), float(dropout_input), float(dropout_features)
        self.feature_grad_mult = float(feature_grad_mult)

    @classmethod
    def from_checkpoint_cfg(cls, cfg):
        if cfg is None:
            return cls()
        m = cfg.get("model", cfg) if isinstance(cfg, dict) else getattr(cfg, "model", cfg)
        get = (lambda k: m.get(k)) if isinstance(m, dict) else (lambda k: getattr(m, k, None))
        return cls(**{k: get(k) for k in cls.FIELDS if get(k) is not None})

    def any_dropout(self):
        return any(getattr(self, k) > 0 for k in self.FIELDS[:6])

    def __repr__(self):
        return "XlsrTrainCfg(%s)" % ", ".join("%s=%g" % (k, getattr(self, k)) for k in self.FIELDS)


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


class XlsrFineTuner(XlsrFrontend):
    """Front-end with a TRAINABLE transformer encoder (24 pre-LN layers + final LayerNorm); the conv feature extractor,
    projection and positional conv stay frozen this round (fairseq's ``feature_grad_mult = 0`` style fine-tuning).

    f32 master weights / gradients live in two flat buffers (one Adam tensor, one all-reduce stream); every step the GEMM
    operands are refreshed from them: W as bf16 (forward) and W^T as bf16 (input gradients).  Weight gradients are
    dW = dY^T.X straight from the row-major bf16 dY / X (occ_gemm_tn: LDS-DMA staging, transposing LDS reads, bf16 MFMA).
    Saved for backward per layer: the f32 residual stream before each LayerNorm, the LayerNorm outputs, qkv, the attention
    output and its log-sum-exp, the pre-GELU and post-GELU FFN activations (about 230 MB per layer at B=32)."""

    def __init__(self, params, cfg, device="cuda"):
        super().__init__(params, cfg, device=device, dtype=torch.bfloat16)
        D, Fd, n = cfg.dim, cfg.ffn, cfg.layers
        if D // cfg.heads not in (64, 80):
            raise OccError("XlsrFineTuner needs head_dim 64 or 80 (attention backward kernel)")
        self.tshapes = []
        for i in range(n):
            self.tshapes += [("l%d.qkv.w" % i, (3 * D, D)), ("l%d.qkv.b" % i, (3 * D,)), ("l%d.o.w" % i, (D, D)), ("l%d.o.b" % i, (D,)),
                             ("l%d.ln1.g" % i, (D,)), ("l%d.ln1.b" % i, (D,)), ("l%d.fc1.w" % i, (Fd, D)), ("l%d.fc1.b" % i, (Fd,)),
                             ("l%d.fc2.w" % i, (D, Fd)), ("l%d.fc2.b" % i, (D,)), ("l%d.ln2.g" % i, (D,)), ("l%d.ln2.b" % i, (D,))]
        self.tshapes += [("enc_ln.g", (D,)), ("enc_ln.b", (D,))]
        self.tshapes += self._extra_shapes()
        off, self.tslots = 0, {}
        for name, shp in self.tshapes:
            nel = 1
            for d in shp:
                nel *= d
            self.tslots[name] = (off, shp, nel)
            off += (nel + 7) // 8 * 8                   # 16-byte aligned slots in the bf16 mirror too (GEMM operands)
        self.P = torch.empty(off, device=self.device, dtype=torch.float32)
        self.G = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.grads_cleared = True
        # bf16 mirror of P: the forward GEMM operands are views into it; the optimizer writes it in the same pass as P (occ_adam_multi
        # bf16_copies), so no cast kernel re-reads the f32 masters every step
        self.Wb = torch.zeros(off, device=self.device, dtype=torch.bfloat16)
        self.mp = {k: self.P[o:o + nel].view(shp) for k, (o, shp, nel) in self.tslots.items()}
        self.mg = {k: self.G[o:o + nel].view(shp) for k, (o, shp, nel) in self.tslots.items()}
        self._load_master(params)
        self.wT = {}
        self._alloc_operands()
        self.refresh_operands()
        self.ctx = None
        # fairseq train-mode behaviour (XlsrTrainCfg; all zero by default): keep-masks are drawn with Philox (drop_seed, step, site) and
        # kept for backward; tests inject them through ``inject_masks`` / ``inject_keep`` to drive the oracle with the same masks
        self.train_cfg = XlsrTrainCfg()
        self.dropout_active = True
        self.drop_seed, self.drop_step = 0, 0
        self.inject_masks, self.inject_keep = None, None
        self.masks, self.keep = {}, None
        self._f8_pre = None                                      # (kind, site) of an fp8 operand a producing kernel has already written
        # partial-sum finalizes of the backward pass in one launch (see _fin_site); OCC_DEFER_FINALIZE=0: one finalize launch per producer
        self.defer_finalize = os.environ.get("OCC_DEFER_FINALIZE", "1") != "0"
        # the paired weight-gradient launches of the transformer layers on a second stream beside the input-gradient chain (see backward):
        # measured +-0 inside the training step, so it is an option, off by default
        self.overlap_wgrad = os.environ.get("OCC_WGRAD_STREAM", "0") == "1"
        self._wg_stream = None
        self._fin_bufs, self._fin_tags, self._fin, self._fin_build = {}, {}, {}, ops.FinalizeBatch()
        self.seed_layerdrop(0)

    def seed_layerdrop(self, seed):
        """Layerdrop decisions come from a per-trainer generator (fairseq draws np.random.random() per layer from the global stream):
        reproducible from (seed, rank), independent across data-parallel ranks."""
        import numpy as np
        self._ld_rng = np.random.RandomState((int(seed) * 2654435761 + 12345) % (2 ** 32))

    def _extra_shapes(self):
        return []

    def _encoder_names(self):
        return [n for n, _ in self.tshapes if n.startswith("l") and n[1].isdigit() or n.startswith("enc_ln")]

    def _alloc_operands(self):
        for name in self._encoder_names():
            shp = self.tslots[name][1]
            if name.endswith(".w"):
                self.wT[name] = torch.empty(shp[1], shp[0], device=self.device, dtype=torch.bfloat16)
                o, _, nel = self.tslots[name]
                self.w[name] = self.Wb[o:o + nel].view(shp)
            else:
                self.w[name] = self.mp[name]                    # biases / LayerNorm affine are used in f32 directly

    def _load_master(self, p):
        with torch.no_grad():
            for i in range(self.cfg.layers):
                pre, a = "encoder.layers.%d" % i, "encoder.layers.%d.self_attn." % i
                f = lambda t: t.detach().to(self.device, torch.float32)
                self.mp["l%d.qkv.w" % i].copy_(torch.cat([f(p[a + "q_proj.weight"]), f(p[a + "k_proj.weight"]), f(p[a + "v_proj.weight"])], 0))
                self.mp["l%d.qkv.b" % i].copy_(torch.cat([f(p[a + "q_proj.bias"]), f(p[a + "k_proj.bias"]), f(p[a + "v_proj.bias"])], 0))
                self.mp["l%d.o.w" % i].copy_(f(p[a + "out_proj.weight"])); self.mp["l%d.o.b" % i].copy_(f(p[a + "out_proj.bias"]))
                self.mp["l%d.ln1.g" % i].copy_(f(p[pre + ".self_attn_layer_norm.weight"])); self.mp["l%d.ln1.b" % i].copy_(f(p[pre + ".self_attn_layer_norm.bias"]))
                self.mp["l%d.fc1.w" % i].copy_(f(p[pre + ".fc1.weight"])); self.mp["l%d.fc1.b" % i].copy_(f(p[pre + ".fc1.bias"]))
                self.mp["l%d.fc2.w" % i].copy_(f(p[pre + ".fc2.weight"])); self.mp["l%d.fc2.b" % i].copy_(f(p[pre + ".fc2.bias"]))
                self.mp["l%d.ln2.g" % i].copy_(f(p[pre + ".final_layer_norm.weight"])); self.mp["l%d.ln2.b" % i].copy_(f(p[pre + ".final_layer_norm.bias"]))
            self.mp["enc_ln.g"].copy_(p["encoder.layer_norm.weight"].detach().to(self.device, torch.float32))
            self.mp["enc_ln.b"].copy_(p["encoder.layer_norm.bias"].detach().to(self.device, torch.float32))

    def refresh_operands(self, cast=True):
        """GEMM operands from the f32 masters: the bf16 mirror Wb (cast=True; after an optimizer step that wrote Wb itself, cast=False)
        and the transposed copies W^T for the input-gradient GEMMs."""
        from .._lib import check, lib, ptr, stream_ptr
        if cast:
            check(lib().occ_cast(ptr(self.P), OCC_F32, ptr(self.Wb), OCC_BF16_CODE, self.P.numel(), stream_ptr()), "occ_cast")
        if getattr(self, "_tbatch", None) is None:             # every W^T operand of the step in one launch, read from the bf16 mirror
            self._tbatch = ops.TransposeBatch()
            self._transpose_jobs(self._tbatch)
        self._tbatch.run()
        if getattr(self, "fp8", False):
            self._fp8_weights()
            if self.ctx is None and self.drop_step > 0:
                self.f8["warm"] = False                          # a whole step has been seen: every site has a measured |max|

    def _transpose_jobs(self, tb):
        for name in self._encoder_names():
            shp = self.tslots[name][1]
            if name.endswith(".w"):
                tb.add(self.w[name], self.wT[name], shp[0], shp[1], ld_src=shp[1], ld_dst=shp[0])

    def ref_views(self, flat):
        """{fairseq name: VIEW of ``flat`` (self.P / self.G / same-sized buffer) in fairseq's shape}: q | k | v are row blocks of the
        fused [3D, D] slot."""
        v = lambda n: flat[self.tslots[n][0]: self.tslots[n][0] + self.tslots[n][2]].view(self.tslots[n][1])
        out, D = {}, self.cfg.dim
        for i in range(self.cfg.layers):
            pre, a = "encoder.layers.%d" % i, "encoder.layers.%d.self_attn." % i
            w, b = v("l%d.qkv.w" % i), v("l%d.qkv.b" % i)
            for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
                out[a + n + ".weight"] = w[j * D:(j + 1) * D]; out[a + n + ".bias"] = b[j * D:(j + 1) * D]
            out[a + "out_proj.weight"] = v("l%d.o.w" % i); out[a + "out_proj.bias"] = v("l%d.o.b" % i)
            out[pre + ".self_attn_layer_norm.weight"] = v("l%d.ln1.g" % i); out[pre + ".self_attn_layer_norm.bias"] = v("l%d.ln1.b" % i)
            out[pre + ".fc1.weight"] = v("l%d.fc1.w" % i); out[pre + ".fc1.bias"] = v("l%d.fc1.b" % i)
            out[pre + ".fc2.weight"] = v("l%d.fc2.w" % i); out[pre + ".fc2.bias"] = v("l%d.fc2.b" % i)
            out[pre + ".final_layer_norm.weight"] = v("l%d.ln2.g" % i); out[pre + ".final_layer_norm.bias"] = v("l%d.ln2.b" % i)
        out["encoder.layer_norm.weight"] = v("enc_ln.g"); out["encoder.layer_norm.bias"] = v("enc_ln.b")
        return out

    def sync_operands(self):
        """Re-derive the GEMM operands (bf16 mirror, W^T copies, weight-normed positional kernel) if anything wrote the f32 masters
        through torch since the last refresh -- an ``optimizer.step()`` on the registered Parameters, a ``load_state_dict`` -- detected by
        the version counter the Parameter views share with ``P``.  (OcTrainer's fused Adam kernel writes the mirror itself and calls
        ``refresh_operands`` directly.)"""
        if self.P._version != getattr(self, "_synced_version", -1):
            self.refresh_operands(cast=True)
            self._synced_version = self.P._version

    def export_params(self):
        """Trainable tensors back under their fairseq names (q/k/v split again)."""
        out, D = {}, self.cfg.dim
        for i in range(self.cfg.layers):
            pre, a = "encoder.layers.%d" % i, "encoder.layers.%d.self_attn." % i
            w, b = self.mp["l%d.qkv.w" % i], self.mp["l%d.qkv.b" % i]
            for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
                out[a + n + ".weight"] = w[j * D:(j + 1) * D].clone(); out[a + n + ".bias"] = b[j * D:(j + 1) * D].clone()
            out[a + "out_proj.weight"] = self.mp["l%d.o.w" % i].clone(); out[a + "out_proj.bias"] = self.mp["l%d.o.b" % i].clone()
            out[pre + ".self_attn_layer_norm.weight"] = self.mp["l%d.ln1.g" % i].clone(); out[pre + ".self_attn_layer_norm.bias"] = self.mp["l%d.ln1.b" % i].clone()
            out[pre + ".fc1.weight"] = self.mp["l%d.fc1.w" % i].clone(); out[pre + ".fc1.bias"] = self.mp["l%d.fc1.b" % i].clone()
            out[pre + ".fc2.weight"] = self.mp["l%d.fc2.w" % i].clone(); out[pre + ".fc2.bias"] = self.mp["l%d.fc2.b" % i].clone()
            out[pre + ".final_layer_norm.weight"] = self.mp["l%d.ln2.g" % i].clone(); out[pre + ".final_layer_norm.bias"] = self.mp["l%d.ln2.b" % i].clone()
        out["encoder.layer_norm.weight"] = self.mp["enc_ln.g"].clone(); out["encoder.layer_norm.bias"] = self.mp["enc_ln.b"].clone()
        return out

    def grad_dict(self):
        out, D = {}, self.cfg.dim
        for i in range(self.cfg.layers):
            pre, a = "encoder.layers.%d" % i, "encoder.layers.%d.self_attn." % i
            w, b = self.mg["l%d.qkv.w" % i], self.mg["l%d.qkv.b" % i]
            for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
                out[a + n + ".weight"] = w[j * D:(j + 1) * D].clone(); out[a + n + ".bias"] = b[j * D:(j + 1) * D].clone()
            out[a + "out_proj.weight"] = self.mg["l%d.o.w" % i].clone(); out[a + "out_proj.bias"] = self.mg["l%d.o.b" % i].clone()
            out[pre + ".self_attn_layer_norm.weight"] = self.mg["l%d.ln1.g" % i].clone(); out[pre + ".self_attn_layer_norm.bias"] = self.mg["l%d.ln1.b" % i].clone()
            out[pre + ".fc1.weight"] = self.mg["l%d.fc1.w" % i].clone(); out[pre + ".fc1.bias"] = self.mg["l%d.fc1.b" % i].clone()
            out[pre + ".fc2.weight"] = self.mg["l%d.fc2.w" % i].clone(); out[pre + ".fc2.bias"] = self.mg["l%d.fc2.b" % i].clone()
            out[pre + ".final_layer_norm.weight"] = self.mg["l%d.ln2.g" % i].clone(); out[pre + ".final_layer_norm.bias"] = self.mg["l%d.ln2.b" % i].clone()
        out["encoder.layer_norm.weight"] = self.mg["enc_ln.g"].clone(); out["encoder.layer_norm.bias"] = self.mg["enc_ln.b"].clone()
        return out

    def _train_ws(self, B, L):
        ws = self._workspace(B, L, slot=0)
        if "tr" not in ws:
            M, D, Fd = ws["M"], self.cfg.dim, self.cfg.ffn
            Mp = (M + 63) // 64 * 64
            bf, f32 = torch.bfloat16, torch.float32
            e = lambda *s, dt=bf: torch.empty(*s, device=self.device, dtype=dt)
            z = lambda *s: torch.zeros(*s, device=self.device, dtype=bf)
            tr = {"Mp": Mp, "layers": []}
            for _ in range(self.cfg.layers):
                tr["layers"].append({"x_in": e(M, D, dt=f32), "h1": e(M, D), "qkv": e(M, 3 * D), "att": e(M, D), "lse": e(ws["M"] // ws["T"] * self.cfg.heads, ws["T"], dt=f32),
                                     "x_mid": e(M, D, dt=f32), "h2": e(M, D), "u": e(M, Fd), "f": e(M, Fd)})
            tr["x_out"] = e(M, D, dt=f32)
            # transposed operand buffers (pad columns stay zero) and gradient activations
            tr["dx"] = e(M, D, dt=f32)
            tr["dxb"] = e(M, D)                                   # bf16 copy of the residual-stream gradient (GEMM operand)
            tr["dxb2"] = e(M, D)                                  # second copy: the two LayerNorm backwards of a layer alternate (weight gradients on a side stream)
            tr["du"], tr["dh"], tr["da"], tr["dqkv"] = e(M, Fd), e(M, D), e(M, D), e(M, 3 * D)
            ws["tr"] = tr
        return ws

    # ---- fp8 transformer GEMMs ---------------------------------------------------------------------------
    _E4 = {"h1": 0, "att": 1, "h2": 2, "f": 3, "qkv.w": 4, "o.w": 5, "fc1.w": 6, "fc2.w": 7}
    _E5 = {"g_qkv": 0, "g_o": 1, "g_fc1": 2, "g_fc2": 3}

    def enable_fp8(self, margin=1.0):
        """Forward and input-gradient GEMMs of the transformer layers on the fp8 MFMA path: activations and weights e4m3, gradients
        e5m2, per-tensor scales.  Activation / gradient sites use DELAYED scaling (this step's scale comes from the previous step's
        |max|; the first step measures |max| before quantising), weights are re-quantised from their bf16 mirrors with their current
        |max| after every optimizer step.  Weight gradients, attention, LayerNorm and the conv / projection prefix stay bf16."""
        from .._lib import OCC_FP8_E4M3, OCC_FP8_E5M2
        cfg, dev = self.cfg, self.device
        n = cfg.layers
        z = lambda k: torch.zeros(k, device=dev, dtype=torch.float32)
        self.f8 = {"e4": OCC_FP8_E4M3, "e5": OCC_FP8_E5M2, "margin": margin, "warm": True,
                   "amax4": z(n * 8), "scale4": torch.ones(n * 8, device=dev), "inv4": torch.ones(n * 8, device=dev),
                   "amax5": z(n * 4), "scale5": torch.ones(n * 4, device=dev), "inv5": torch.ones(n * 4, device=dev), "wq": {}, "wtq": {}, "qa": None, "qg": None, "qa1": None, "qg1": None}
        for i in range(n):
            for wn in ("qkv.w", "o.w", "fc1.w", "fc2.w"):
                name = "l%d.%s" % (i, wn)
                self.f8["wq"][name] = torch.empty(self.w[name].shape, device=dev, dtype=torch.uint8)
                self.f8["wtq"][name] = torch.empty(self.wT[name].shape, device=dev, dtype=torch.uint8)
        self.fp8 = True
        self._fp8_weights()

    def _fp8_weights(self):
        """Weights: |max| of the bf16 mirror now -> scale -> e4m3 copies of W and W^T (same scale)."""
        f8 = self.f8
        if "w_amax" not in f8:                                   # two launches for all weights: |max| of every W, then W and W^T quantised
            f8["w_amax"], f8["w_quant"] = ops.Fp8Batch(f8["e4"]), ops.Fp8Batch(f8["e4"])
            for name in f8["wq"]:
                i, wn = int(name[1:name.index(".")]), name[name.index(".") + 1:]
                k = i * 8 + self._E4[wn]
                f8["w_amax"].add(self.w[name], amax=f8["amax4"][k:k + 1])
                f8["w_quant"].add(self.w[name], f8["wq"][name], scale=f8["scale4"][k:k + 1])
                f8["w_quant"].add(self.wT[name], f8["wtq"][name], scale=f8["scale4"][k:k + 1])
        f8["w_amax"].run()
        # one pass over every e4m3 / e5m2 site: the activation and gradient sites pick up this step's |max| for the next step
        ops.fp8_update_scales(f8["amax4"], f8["scale4"], f8["inv4"], f8["e4"], f8["margin"])
        ops.fp8_update_scales(f8["amax5"], f8["scale5"], f8["inv5"], f8["e5"], f8["margin"])
        f8["w_quant"].run()

    def _fp8_q(self, x, kind, k):
        """x (bf16 / f32) -> fp8 scratch; kind 4 = e4m3 activation site, 5 = e5m2 gradient site; returns (buffer, inverse-scale scalar)."""
        f8 = self.f8
        fmt, am, sc, inv = (f8["e4"], f8["amax4"], f8["scale4"], f8["inv4"]) if kind == 4 else (f8["e5"], f8["amax5"], f8["scale5"], f8["inv5"])
        key = "qa" if kind == 4 else "qg"
        if f8[key] is None or f8[key].numel() < x.numel():
            f8[key] = torch.empty(x.numel(), device=self.device, dtype=torch.uint8)
        q = f8[key][: x.numel()]
        if f8["warm"]:                                           # no history yet: measure, derive the scale, then quantise (current scaling)
            ops.fp8_amax(x, am[k:k + 1])
            ops.fp8_update_scales(am[k:k + 1], sc[k:k + 1], inv[k:k + 1], fmt, f8["margin"])
        ops.fp8_quantize(x, q, fmt, scale=sc[k:k + 1], amax=am[k:k + 1])
        return q, inv[k:k + 1]

    def _lin(self, i, wn, a, site, M, N, K, C, c_map, c_dtype, f8_next=None, **kw):
        """Forward Linear of layer i (weight "l<i>.<wn>", input a [M,K] bf16): bf16 MFMA, or fp8 with the site's delayed scale.
        f8_next (fp8 path, delayed scales known): the activation site whose e4m3 operand is this GEMM's bf16 result -- the epilogue then
        writes it (other operand buffer) and the consumer skips its quantisation pass."""
        name = "l%d.%s" % (i, wn)
        if not getattr(self, "fp8", False):
            ops.gemm_raw(M, N, K, a, rowmap(M, 0, K), self.w[name], K, C, c_map, c_dtype, OCC_BF16_CODE, **kw)
            return
        k4 = i * 8 + self._E4[site]
        if self._f8_pre is not None and self._f8_pre[:2] == (4, k4):      # the producing kernel already wrote the e4m3 operand
            slot = self._f8_pre[2]
            q, inv_a = self._f8_buf(4, a.numel(), slot), self.f8["inv4"][k4:k4 + 1]
        else:
            slot = 0
            q, inv_a = self._fp8_q(a, 4, k4)
        self._f8_pre = None
        kw_ = i * 8 + self._E4[wn]
        if f8_next is not None and not self.f8["warm"] and M >= 256 and N >= 256:
            kn = i * 8 + self._E4[f8_next]
            kw["c_f8"] = (self._f8_buf(4, M * N, 1 - slot), self.f8["scale4"][kn:kn + 1], self.f8["amax4"][kn:kn + 1], self.f8["e4"])
            self._f8_pre = (4, kn, 1 - slot)
        ops.gemm_raw(M, N, K, q, rowmap(M, 0, K), self.f8["wq"][name], K, C, c_map, c_dtype, self.f8["e4"], a_dequant=inv_a, w_dequant=self.f8["inv4"][kw_:kw_ + 1], **kw)

    def _dgrad(self, i, wn, dy, gsite, M, N, K, C, f8_next=None, colsum=None, **kw):
        """Input gradient dX [M,N] = dY [M,K] . W (through W^T [N,K]) of layer i's weight wn.  f8_next: the gradient site whose e5m2 operand
        is this GEMM's bf16 result (as in _lin)."""
        name = "l%d.%s" % (i, wn)
        if colsum is not None:                                   # (the caller has checked that this launch takes the 256-row kernel)
            kw["c_colsum"] = colsum
        if not getattr(self, "fp8", False):
            ops.gemm_raw(M, N, K, dy, rowmap(M, 0, K), self.wT[name], K, C, rowmap(M, 0, N), OCC_BF16_CODE, OCC_BF16_CODE, **kw)
            return
        k5 = i * 4 + self._E5[gsite]
        if self._f8_pre is not None and self._f8_pre[:2] == (5, k5):      # the producing kernel already wrote the e5m2 operand
            slot = self._f8_pre[2]
            q, inv_g = self._f8_buf(5, dy.numel(), slot), self.f8["inv5"][k5:k5 + 1]
        else:
            slot = 0
            q, inv_g = self._fp8_q(dy, 5, k5)
        self._f8_pre = None
        kw_ = i * 8 + self._E4[wn]
        if f8_next is not None and not self.f8["warm"] and M >= 256 and N >= 256:
            kn = i * 4 + self._E5[f8_next]
            kw["c_f8"] = (self._f8_buf(5, M * N, 1 - slot), self.f8["scale5"][kn:kn + 1], self.f8["amax5"][kn:kn + 1], self.f8["e5"])
            self._f8_pre = (5, kn, 1 - slot)
        ops.gemm_raw(M, N, K, q, rowmap(M, 0, K), self.f8["wtq"][name], K, C, rowmap(M, 0, N), OCC_BF16_CODE, self.f8["e5"], a_dequant=inv_g,
                     w_dequant=self.f8["inv4"][kw_:kw_ + 1], **kw)

    def _f8_buf(self, kind, numel, slot=0):
        """fp8 operand scratch: two buffers per kind (4 = e4m3 activations, 5 = e5m2 gradients) -- a GEMM that reads its fp8 operand from one
        can write the fp8 copy of its own result (the next GEMM's operand) into the other."""
        key = ("qa" if kind == 4 else "qg") + ("" if slot == 0 else "1")
        if self.f8.get(key) is None or self.f8[key].numel() < numel:
            self.f8[key] = torch.empty(numel, device=self.device, dtype=torch.uint8)
        return self.f8[key][:numel]

    def _ln_fwd(self, i, x, ln, out, site):
        """LayerNorm `ln` ("ln1" / "ln2") of layer i into the bf16 operand `out`; on the fp8 path (delayed scales known) the same kernel also
        writes the e4m3 operand of the GEMM that follows (site "h1" / "h2"): no stand-alone quantisation pass."""
        g, b = self.w["l%d.%s.g" % (i, ln)], self.w["l%d.%s.b" % (i, ln)]
        if getattr(self, "fp8", False) and not self.f8["warm"]:
            k4 = i * 8 + self._E4[site]
            ops.layernorm_fp8(x, g, b, out, self._f8_buf(4, out.numel()), self.f8["scale4"][k4:k4 + 1], self.f8["amax4"][k4:k4 + 1])
            self._f8_pre = (4, k4, 0)
        else:
            ops.layernorm(x, g, b, out=out)

    def _ln_bwd(self, dy, x, gname, dres, dx, dxb, bias_name=None, f8_site=None):
        """LayerNorm backward into dx (f32) + dxb (bf16).  Where the gradient it produces is the output gradient of a Linear (x = residual +
        Linear(.): out-proj / fc2), the same kernel also sums its columns into that Linear's bias gradient (bias_name) and, on the fp8 path,
        writes the e5m2 operand of that Linear's input-gradient GEMM (f8_site = index into the gradient-site scales).  Returns True when the
        bias gradient was taken care of."""
        M, C = dx.shape
        f8 = getattr(self, "fp8", False) and f8_site is not None and not self.f8["warm"]
        if (bias_name is not None or f8) and M >= 2048 and C <= 1536:
            q = self._f8_buf(5, M * C) if f8 else None
            dbias = self.mg[bias_name] if bias_name is not None else None
            part = self._fin_site("ln." + gname, 768 * C)
            ops.layernorm_bwd_fused(dy, x, self.w[gname + ".g"], dres, dx, self.mg[gname + ".g"], self.mg[gname + ".b"], dxb, dbias=dbias, dx_f8=q,
                                    f8_scale=self.f8["scale5"][f8_site:f8_site + 1] if f8 else None, f8_amax=self.f8["amax5"][f8_site:f8_site + 1] if f8 else None,
                                    defer=part)
            if part is not None:
                self._fin_build.add_ln(part, M, C, self.mg[gname + ".g"], self.mg[gname + ".b"], dbias)
            if f8:
                self._f8_pre = (5, f8_site, 0)
            return bias_name is not None
        ops.layernorm_bwd(dy, x, self.w[gname + ".g"], dres, dx, self.mg[gname + ".g"], self.mg[gname + ".b"], dx_bf16=dxb)
        return False

    # ---- deferred finalizes ----------------------------------------------------------------------------------------------
    # The fused LayerNorm backward, the GELU' epilogue's column sums and the attention backward's bias sums leave per-workgroup partial sums;
    # each site keeps them in a buffer of its own and ONE occ_finalize_batch launch adds them into the gradients -- at the end of the
    # backward pass, or per layer when a data-parallel caller wants the layer's gradients final for its all-reduce (grad_ready).
    def _fin_site(self, site, nfloats, zero_tag=None):
        """This site's partial-sum buffer.  zero_tag (the GELU' column sums): the producer may write FEWER partial rows than the job sums --
        occ_gemm picks 208-, 224- or 256-row tiles, the job counts rows for the smallest -- so the buffer starts as zeros and is cleared again
        whenever the launch shape (zero_tag) changes: rows the kernel of one shape never writes then always read as zero."""
        if not self.defer_finalize:
            return None
        buf = self._fin_bufs.get(site)
        if buf is None or buf.numel() < nfloats:
            buf = self._fin_bufs[site] = (torch.zeros if zero_tag is not None else torch.empty)(nfloats, device=self.device, dtype=torch.float32)
            self._fin_tags[site] = zero_tag
        elif zero_tag is not None and self._fin_tags.get(site) != zero_tag:
            buf.zero_()
            self._fin_tags[site] = zero_tag
        return buf

    def _fin_flush(self, key):
        if not self._fin_build.jobs:
            return
        fb = self._fin.get(key)
        if fb is None:
            fb = self._fin[key] = ops.FinalizeBatch()
        fb.jobs, self._fin_build.jobs = self._fin_build.jobs, []
        fb._refs, self._fin_build._refs = self._fin_build._refs, {}
        fb.run()

    # ---- train-mode dropouts ---------------------------------------------------------------------------------------------
    def _p(self, field):
        return getattr(self.train_cfg, field) if self.dropout_active else 0.0

    def _mask(self, site, like):
        """Keep-mask (u8, like.numel()) of a dropout site: injected by a test, or a fresh buffer that occ_dropout_ex fills (generate=1)."""
        if self.inject_masks is not None:
            m = self.inject_masks[site].to(self.device, torch.uint8).contiguous().view(-1)
            self.masks[site] = m
            return m, False
        m = self.masks.get(site)
        if m is None or m.numel() != like.numel():
            m = self.masks[site] = torch.empty(like.numel(), device=self.device, dtype=torch.uint8)
        return m, True

    def _drop_fwd(self, site, x, y, p, residual=None):
        """y = residual + dropout_p(x) at `site` (train mode)."""
        m, gen = self._mask(site, x)
        import zlib
        sid = (self.drop_step << 12) + (zlib.crc32(site.encode()) & 0xfff)
        ops.dropout_ex(x, y, m, p, seed=self.drop_seed, stream_id=sid, generate=gen, residual=residual)

    def _drop_bwd(self, site, dy, dx, p):
        ops.dropout_ex(dy, dx, self.masks[site], p, generate=False)

    def _att_keep(self, i, B, T):
        """Keep-mask u8 [B*H, T, Tp] (Tp = T rounded up to 4) of layer i's attention probabilities: injected as [B,H,T,T] by a test, or
        drawn on the device (Philox keyed by drop_seed, step and the site id, like every other dropout site)."""
        import zlib
        site, H = "l%d.att" % i, self.cfg.heads
        Tp = (T + 3) // 4 * 4
        if self.inject_masks is not None:
            m = torch.zeros(B * H, T, Tp, device=self.device, dtype=torch.uint8)
            m[:, :, :T] = self.inject_masks[site].to(self.device, torch.uint8).reshape(B * H, T, T)
            self.masks[site] = m
            return m
        m = self.masks.get(site)
        if m is None or m.shape != (B * H, T, Tp):
            m = self.masks[site] = torch.empty(B * H, T, Tp, device=self.device, dtype=torch.uint8)
        sid = (self.drop_step << 12) + (zlib.crc32(site.encode()) & 0xfff)
        ops.dropout_mask(m, self._p("attention_dropout"), seed=self.drop_seed, stream_id=sid)
        return m

    def forward_train(self, wav):
        """wav f32 [B,L] -> features f32 [B,T,dim]; keeps the tape for backward()."""
        cfg, w = self.cfg, self.w
        B, L = wav.shape
        ws = self._train_ws(B, L)
        tr = ws["tr"]
        T, M, D, Fd = ws["T"], ws["M"], cfg.dim, cfg.ffn
        p_res, p_act, p_ld, p_att = self._p("dropout"), self._p("activation_dropout"), self._p("encoder_layerdrop"), self._p("attention_dropout")
        self.drop_step += 1
        if self.inject_keep is not None:
            self.keep = [bool(k) for k in self.inject_keep]
        elif p_ld > 0:                                           # fairseq: np.random.random() > layerdrop keeps the layer
            self.keep = [bool(self._ld_rng.random_sample() > p_ld) for _ in range(cfg.layers)]
        else:
            self.keep = [True] * cfg.layers
        if (p_res > 0 or p_act > 0) and "y" not in tr:
            tr["y"] = torch.empty(M, D, device=self.device, dtype=torch.float32)       # branch output before its dropout
            tr["dyb"] = torch.empty(M, D, device=self.device, dtype=torch.bfloat16)    # dropped gradient of a branch output
        # The f32 residual stream is never copied: every layer keeps its input (x_in) and its middle state (x_mid) for backward, so
        # out-proj reads x_in and writes x_mid, fc2 reads x_mid and writes the NEXT layer's x_in (the last one writes x_out), and the
        # prefix writes layer 0's x_in directly.
        with torch.no_grad():
            self._frozen_prefix(wav, ws, x_out=tr["layers"][0]["x_in"])       # conv stack .. positional conv -> f32 residual stream
        code = OCC_BF16_CODE
        xmap, hd = rowmap(M, 0, D), D // cfg.heads
        if p_res > 0:                                            # TransformerEncoder.extract_features: x = F.dropout(x + pos_conv(x), p=dropout)
            x0 = tr["layers"][0]["x_in"]
            self._drop_fwd("enc", x0, x0, p_res)
        for i in range(cfg.layers):
            s = tr["layers"][i]
            x_in, x_mid = s["x_in"], s["x_mid"]
            x_next = tr["layers"][i + 1]["x_in"] if i + 1 < cfg.layers else tr["x_out"]
            if not self.keep[i]:                                 # layerdrop: the layer is skipped, the residual stream passes through
                x_next.copy_(x_in)
                continue
            self._ln_fwd(i, x_in, "ln1", s["h1"], "h1")
            self._lin(i, "qkv.w", s["h1"], "h1", M, 3 * D, D, s["qkv"], rowmap(M, 0, 3 * D), code, bias=w["l%d.qkv.b" % i])
            if p_att > 0:                                        # MultiheadAttention: dropout on the attention probabilities (kept for backward)
                ops.attention_dropout(s["qkv"], B, T, cfg.heads, hd, hd ** -0.5, self._att_keep(i, B, T), p_att, out=s["att"], lse=s["lse"])
            else:
                ops.attention(s["qkv"], B, T, cfg.heads, hd, hd ** -0.5, out=s["att"], lse=s["lse"])
            if p_res > 0:                                        # x = residual + dropout1(self_attn(LN(x)))
                self._lin(i, "o.w", s["att"], "att", M, D, D, tr["y"], xmap, OCC_F32, bias=w["l%d.o.b" % i])
                self._drop_fwd("l%d.d1" % i, tr["y"], x_mid, p_res, residual=x_in)
            else:
                self._lin(i, "o.w", s["att"], "att", M, D, D, x_mid, xmap, OCC_F32, bias=w["l%d.o.b" % i], R=x_in, r_map=xmap, r_dtype=OCC_F32)
            self._ln_fwd(i, x_mid, "ln2", s["h2"], "h2")
            self._lin(i, "fc1.w", s["h2"], "h2", M, Fd, D, s["f"], rowmap(M, 0, Fd), code, bias=w["l%d.fc1.b" % i], act=ACT_GELU_KEEP_GRAD if KEEP_GELU_GRAD else ACT_GELU, aux=s["u"],
                      f8_next="f" if p_act == 0 else None)
            if p_act > 0:                                        # dropout2 on the activation
                self._drop_fwd("l%d.act" % i, s["f"], s["f"], p_act)
            if p_res > 0:                                        # x = residual + dropout3(fc2(.))
                self._lin(i, "fc2.w", s["f"], "f", M, D, Fd, tr["y"], xmap, OCC_F32, bias=w["l%d.fc2.b" % i])
                self._drop_fwd("l%d.d3" % i, tr["y"], x_next, p_res, residual=x_mid)
            else:
                self._lin(i, "fc2.w", s["f"], "f", M, D, Fd, x_next, xmap, OCC_F32, bias=w["l%d.fc2.b" % i], R=x_mid, r_map=xmap, r_dtype=OCC_F32)
        out = torch.empty(B, T, D, device=self.device, dtype=torch.float32)
        ops.layernorm(tr["x_out"], w["enc_ln.g"], w["enc_ln.b"], out=out.view(M, D))
        self.ctx = (B, L)
        return out

    def _frozen_prefix(self, wav, ws, x_out=None):
        """Conv feature extractor, LayerNorm, projection, positional conv (frozen): fills x_out (default ws["x"])."""
        x_out = ws["x"] if x_out is None else x_out
        cfg, w, dt = self.cfg, self.w, self.dtype
        wav = wav.to(self.device, torch.float32).contiguous()
        B, L = wav.shape
        Ts, T, M = ws["Ts"], ws["T"], ws["M"]
        code = dtype_code(ws["h"])
        D = cfg.dim
        cur, nxt = ws["cA"], ws["cB"]
        ops.conv0_ln_gelu(wav, w["c0.w"], w["c0.b"], w["c0.g"], w["c0.be"], 10, 5, dt, out=cur[: B * Ts[0] * 512].view(B, Ts[0], 512))
        Tin = Ts[0]
        for i in range(1, 7):
            _, k, s = CONV_LAYERS[i]
            Tout = Ts[i]
            o = nxt[: B * Tout * 512].view(B * Tout, 512)
            ops.gemm_raw(B * Tout, 512, k * 512, cur, rowmap(Tout, Tin * 512, s * 512), w["c%d.w" % i], k * 512, o, rowmap(B * Tout, 0, 512), code, code, bias=w["c%d.b" % i])
            ops.layernorm(o, w["c%d.g" % i], w["c%d.be" % i], gelu=True, out=o)
            cur, nxt = nxt, cur
            Tin = Tout
        ops.layernorm(cur[: M * 512].view(M, 512), w["ln.g"], w["ln.b"], out=ws["feat"])
        xpad = ws["xpad"]
        Tp, half = T + cfg.pos_k, cfg.pos_k // 2
        inner = xpad.data_ptr() + half * D * xpad.element_size()
        pmap = rowmap(T, Tp * D, D)
        ops.gemm_raw(M, D, 512, ws["feat"], rowmap(M, 0, 512), w["proj.w"], 512, inner, pmap, code, code, bias=w["proj.b"])
        G = cfg.pos_groups
        cg = D // G
        ops.gemm_raw(M, cg, cfg.pos_k * cg, xpad, pmap, w["pos.w"], cfg.pos_k * cg, x_out, rowmap(M, 0, D), OCC_F32, code, bias=w["pos.b"], act=ACT_GELU,
                     R=inner, r_map=pmap, r_dtype=code, a_seg=(cfg.pos_k, cg, D), groups=(G, cg, cg * cfg.pos_k * cg, cg))

    def zero_grad(self):
        from .. import backend_ops as K
        K.fill(self.G, 0.0)
        self.grads_cleared = True            # until the next backward(): the weight-gradient kernels may store instead of accumulate

    def _wgrad(self, dy, x, N, Kd, M, gname, bias_name, c_is_zero=False):
        """G[gname] [N,Kd] += dy^T x ; G[bias] += colsum(dy).  dy [M,N], x [M,Kd], both bf16 as they lie in memory: occ_gemm_tn's
        LDS-DMA / transposing-read kernel needs no transposed copies."""
        from .. import backend_ops as K
        K.gemm_tn(M, N, Kd, dy, rowmap(M, 0, N), x, rowmap(M, 0, Kd), self.mg[gname], Kd, colsum_out=self.mg[bias_name], a_bf16=True, b_bf16=True,
                  bf16_mfma=True, c_is_zero=c_is_zero)

    def _wgrad_pair(self, M, a, b):
        """Two weight (+ bias) gradients of one layer, (dy, x, N, Kd, gname, bias_name) each, in one launch (occ_gemm_tn_pair)."""
        from .. import backend_ops as K
        K.gemm_tn_pair(M, *[(N, Kd, dy, rowmap(M, 0, N), x, rowmap(M, 0, Kd), self.mg[g], Kd, self.mg[bn] if bn is not None else None) for dy, x, N, Kd, g, bn in (a, b)],
                       c_is_zero=self.grads_cleared)

    def layer_grad_range(self, i):
        """[lo, hi) of transformer layer i's gradients in the flat buffer self.G (its twelve tensors are contiguous)."""
        lo = self.tslots["l%d.qkv.w" % i][0]
        hi = self.tslots["l%d.qkv.w" % (i + 1)][0] if i + 1 < self.cfg.layers else self.tslots["enc_ln.g"][0]
        return lo, hi

    def backward(self, dfeats, grad_ready=None):
        """dfeats f32 [B,T,dim] (gradient wrt the returned features) -> accumulates into self.G.
        grad_ready(lo, hi): called after each transformer layer (last first) with the range of self.G that is now final, so a
        data-parallel caller can start that slice's all-reduce under the rest of the backward pass."""
        if self.ctx is None:
            raise OccError("backward() needs a preceding forward_train()")
        cfg, w = self.cfg, self.w
        B, L = self.ctx
        ws = self._workspace(B, L, slot=0)
        tr = ws["tr"]
        T, M, D, Fd, Mp = ws["T"], ws["M"], cfg.dim, cfg.ffn, tr["Mp"]
        bfc, hd = OCC_BF16_CODE, D // cfg.heads
        xmap, fmap, qmap = rowmap(M, 0, D), rowmap(M, 0, Fd), rowmap(M, 0, 3 * D)
        dx, dxb = tr["dx"], tr["dxb"]
        p_res, p_act, p_att = self._p("dropout"), self._p("activation_dropout"), self._p("attention_dropout")
        # A LayerNorm backward's output is the output gradient of the Linear below it in the residual chain (the final LayerNorm and every
        # ln1: fc2 of the next kept layer down; ln2: this layer's out-proj): with no residual dropout in between that kernel also produces the
        # Linear's bias gradient and, on the fp8 path, its e5m2 GEMM operand.
        fuse = p_res == 0
        # Weight gradients beside the input-gradient chain.  Every large launch here owns whole CUs (128 KiB of LDS or all registers), and
        # the input-gradient GEMMs leave CUs idle: 228 tiles for 256 CUs at N = 1024, a 0.56-full last round at N = 4096, and their epilogues
        # stall the matrix cores.  The two paired weight-gradient launches of a layer depend only on operands the chain has already
        # produced, so they can go to a second stream and fill those holes (scripts/bench_two_streams.py, GEMMs only: 690 -> 628 us per
        # layer; inside the whole step, with the LayerNorm / attention backward kernels in the chain: +-0.2 ms either way).  Hazards:
        # a pair reads the bf16 residual gradient of ITS LayerNorm backward, du / dqkv and saved activations; the two LayerNorm backwards
        # of a layer therefore write alternate bf16 buffers (A: enc_ln / ln1, B: ln2), and the main stream waits for the FFN pair
        # before ln1 rewrites A (and, one layer on, du) and for the attention pair before the next ln2 / attention backward rewrite B / dqkv.
        # Not with residual / activation / attention dropout (their gradient buffers are reused within the layer), not while a caller
        # wants per-layer gradients (grad_ready: the data-parallel all-reduce), not under stream capture.
        side = None
        if self.overlap_wgrad and grad_ready is None and p_res == 0 and p_act == 0 and p_att == 0 and not torch.cuda.is_current_stream_capturing():
            if self._wg_stream is None:
                self._wg_stream = torch.cuda.Stream(device=self.device)
            side = self._wg_stream
            main = torch.cuda.current_stream()
            side.wait_stream(main)
        dxbA, dxbB = tr["dxb"], (tr["dxb2"] if side is not None else tr["dxb"])
        ffn_done = att_done = None                               # events: the side stream has finished the layer's FFN / attention pair

        def on_side(fn, after_main=True):
            """Run fn (a weight-gradient pair) on the side stream once the main stream has reached this point; returns its completion event."""
            if side is None:
                fn()
                return None
            ready = torch.cuda.Event(); ready.record(main)
            side.wait_event(ready)
            with torch.cuda.stream(side):
                fn()
                done = torch.cuda.Event(); done.record(side)
            return done

        kept = [i for i in range(cfg.layers) if self.keep[i]]
        below = lambda i: max([j for j in kept if j < i], default=None)      # the kept layer whose output gradient ln1-backward of layer i produces
        top = kept[-1] if kept else None
        fc2_bias_done = {}
        done = self._ln_bwd(dfeats.contiguous().view(M, D), tr["x_out"], "enc_ln", None, dx, dxb, bias_name="l%d.fc2.b" % top if fuse and top is not None else None,
                            f8_site=top * 4 + self._E5["g_fc2"] if fuse and top is not None else None)
        if top is not None:
            fc2_bias_done[top] = done
        for i in range(cfg.layers - 1, -1, -1):
            s = tr["layers"][i]
            if not self.keep[i]:                                 # a dropped layer: the gradient passes through, its parameters get none
                if grad_ready is not None:
                    grad_ready(*self.layer_grad_range(i))
                continue
            # ---- FFN: x3 = x_mid + dropout3(fc2(dropout2(gelu(fc1(LN2(x_mid))))))
            dyb = dxbA
            if p_res > 0:
                dyb = tr["dyb"]; self._drop_bwd("l%d.d3" % i, dxbA, dyb, p_res)
            # fc1's bias gradient = column sums of du: from the same epilogue that writes du (large launches, no activation dropout on du)
            fc1_bias_fused = p_act == 0 and M * Fd >= 180 * 65536 and Fd % 8 == 0 and D % 64 == 0
            colsum = None
            if fc1_bias_fused:
                colsum = self.mg["l%d.fc1.b" % i]
                part = self._fin_site("fc1.%d" % i, 2 * ((M + 207) // 208) * Fd, zero_tag=(M, Fd, D, bool(getattr(self, "fp8", False))))
                if part is not None:
                    self._fin_build.add_rows(part, 2 * ((M + 207) // 208), colsum)
                    colsum = (colsum, part)
            self._dgrad(i, "fc2.w", dyb, "g_fc2", M, Fd, D, tr["du"], act=ACT_MUL_AUX if KEEP_GELU_GRAD else ACT_GELU_GRAD, aux=s["u"], f8_next="g_fc1" if p_act == 0 else None,
                        colsum=colsum)
            if p_act > 0:                                        # (elementwise factors commute: mask after GELU')
                self._drop_bwd("l%d.act" % i, tr["du"], tr["du"], p_act)
            # both FFN weight gradients in one launch (dyb is not rewritten before the LayerNorm backward below)
            ffn_done = on_side(lambda dyb=dyb, s=s, i=i, b2=fc2_bias_done.get(i), b1=fc1_bias_fused: self._wgrad_pair(
                M, (dyb, s["f"], D, Fd, "l%d.fc2.w" % i, None if b2 else "l%d.fc2.b" % i), (tr["du"], s["h2"], Fd, D, "l%d.fc1.w" % i, None if b1 else "l%d.fc1.b" % i)))
            self._dgrad(i, "fc1.w", tr["du"], "g_fc1", M, D, Fd, tr["dh"])
            if att_done is not None:                             # the layer above's attention pair still reads buffer B (and dqkv)
                main.wait_event(att_done); att_done = None
            o_bias_done = self._ln_bwd(tr["dh"], s["x_mid"], "l%d.ln2" % i, dx, dx, dxbB, bias_name="l%d.o.b" % i if fuse else None,
                                       f8_site=i * 4 + self._E5["g_o"] if fuse else None)
            # ---- attention: x_mid = x_in + dropout1(out_proj(attn(qkv(LN1(x_in)))))
            dyb = dxbB
            if p_res > 0:
                dyb = tr["dyb"]; self._drop_bwd("l%d.d1" % i, dxbB, dyb, p_res)
            self._dgrad(i, "o.w", dyb, "g_o", M, D, D, tr["da"])
            qkv_bias_done = False
            if p_att > 0:
                ops.attention_bwd_dropout(s["qkv"], s["att"], tr["da"], s["lse"], B, T, cfg.heads, hd, hd ** -0.5, self.masks["l%d.att" % i], p_att, dqkv=tr["dqkv"])
            elif T <= 256:                                       # one key block: the kernel that writes dqkv also sums its columns (the qkv bias gradient)
                part = self._fin_site("att.%d" % i, B * cfg.heads * 3 * hd)
                ops.attention_bwd_bias(s["qkv"], s["att"], tr["da"], s["lse"], B, T, cfg.heads, hd, hd ** -0.5, self.mg["l%d.qkv.b" % i], dqkv=tr["dqkv"], defer=part)
                if part is not None:
                    self._fin_build.add_attention_bias(part, B, cfg.heads, hd, self.mg["l%d.qkv.b" % i])
                qkv_bias_done = True
            else:
                ops.attention_bwd(s["qkv"], s["att"], tr["da"], s["lse"], B, T, cfg.heads, hd, hd ** -0.5, dqkv=tr["dqkv"])
            att_done = on_side(lambda dyb=dyb, s=s, i=i, bo=o_bias_done, bq=qkv_bias_done: self._wgrad_pair(
                M, (dyb, s["att"], D, D, "l%d.o.w" % i, None if bo else "l%d.o.b" % i), (tr["dqkv"], s["h1"], 3 * D, D, "l%d.qkv.w" % i, None if bq else "l%d.qkv.b" % i)))
            self._dgrad(i, "qkv.w", tr["dqkv"], "g_qkv", M, D, 3 * D, tr["dh"])
            j = below(i)
            if ffn_done is not None:                             # this layer's FFN pair still reads buffer A (and du)
                main.wait_event(ffn_done); ffn_done = None
            done = self._ln_bwd(tr["dh"], s["x_in"], "l%d.ln1" % i, dx, dx, dxbA, bias_name="l%d.fc2.b" % j if fuse and j is not None else None,
                                f8_site=j * 4 + self._E5["g_fc2"] if fuse and j is not None else None)
            if j is not None:
                fc2_bias_done[j] = done
            if grad_ready is not None:
                self._fin_flush((M, i))                          # (also holds fc2.b of the layer below: final before that layer's own call)
                grad_ready(*self.layer_grad_range(i))
        if side is not None:
            main.wait_stream(side)                               # every weight gradient is in G (the optimizer and the conv stack's backward follow)
        self._fin_flush((M, "tail", grad_ready is not None))
        if p_res > 0:                                            # the encoder's input dropout
            self._drop_bwd("enc", dx, dx, p_res)
            self._drop_bwd("enc", dxb, dxb, p_res)
        self.ctx = None
        self.grads_cleared = False           # a second backward() without zero_grad() accumulates


class XlsrFullFineTuner(XlsrFineTuner):
    """End-to-end trainable XLS-R: conv feature extractor, LayerNorm, projection, weight-normed positional conv AND the
    transformer encoder (what the reference's optimizer holds, oc_training.py:324).

    Conv blocks 1-6 backward: LayerNorm+GELU backward fused in one row kernel (the pre-LN conv output is kept in bf16); weight
    gradients as bf16 MFMA GEMMs over transposed operands, where the transposed *window* operand is k strided transposes of the
    previous activation; input gradients as plain GEMMs over the zero-padded output gradient -- even input frames take taps
    (2, 0) from two consecutive output rows, odd frames tap 1 (k=3, s=2), or both taps of one row (k=2, s=2).  Block 0 is
    recomputed from the waveform in its backward kernel.  The positional conv differentiates through weight_norm."""

    def _extra_shapes(self):
        cfg = self.cfg
        shp = [("c0.w", (512, 10)), ("c0.b", (512,)), ("c0.g", (512,)), ("c0.be", (512,))]
        for i in range(1, 7):
            k = CONV_LAYERS[i][1]
            shp += [("c%d.w" % i, (512, k, 512)), ("c%d.b" % i, (512,)), ("c%d.g" % i, (512,)), ("c%d.be" % i, (512,))]
        shp += [("ln.g", (512,)), ("ln.b", (512,)), ("proj.w", (cfg.dim, 512)), ("proj.b", (cfg.dim,)),
                ("pos.v", (cfg.dim, cfg.dim // cfg.pos_groups, cfg.pos_k)), ("pos.g", (cfg.pos_k,)), ("pos.b", (cfg.dim,))]
        return shp

    def __init__(self, params, cfg, device="cuda"):
        self._full = True
        super().__init__(params, cfg, device=device)

    # hooks used by XlsrFineTuner.__init__ -------------------------------------------------------------------------
    def _load_master(self, p):
        super()._load_master(p)
        f = lambda t: t.detach().to(self.device, torch.float32)
        with torch.no_grad():
            for i, (c, k, s) in enumerate(CONV_LAYERS):
                pre = "feature_extractor.conv_layers.%d" % i
                w = f(p[pre + ".0.weight"])
                self.mp["c%d.w" % i].copy_(w.reshape(512, -1) if i == 0 else w.permute(0, 2, 1))
                self.mp["c%d.b" % i].copy_(f(p[pre + ".0.bias"]))
                self.mp["c%d.g" % i].copy_(f(p[pre + ".2.1.weight"])); self.mp["c%d.be" % i].copy_(f(p[pre + ".2.1.bias"]))
            self.mp["ln.g"].copy_(f(p["layer_norm.weight"])); self.mp["ln.b"].copy_(f(p["layer_norm.bias"]))
            self.mp["proj.w"].copy_(f(p["post_extract_proj.weight"])); self.mp["proj.b"].copy_(f(p["post_extract_proj.bias"]))
            self.mp["pos.v"].copy_(f(p["encoder.pos_conv.0.weight_v"])); self.mp["pos.g"].copy_(f(p["encoder.pos_conv.0.weight_g"]).reshape(-1))
            self.mp["pos.b"].copy_(f(p["encoder.pos_conv.0.bias"]))

    def _alloc_operands(self):
        super()._alloc_operands()
        cfg, dev, bf = self.cfg, self.device, torch.bfloat16
        for name in ("c0.w", "c0.b", "c0.g", "c0.be", "ln.g", "ln.b", "proj.b", "pos.b"):
            self.w[name] = self.mp[name]
        for i in range(1, 7):
            k = CONV_LAYERS[i][1]
            for n in ("b", "g", "be"):
                self.w["c%d.%s" % (i, n)] = self.mp["c%d.%s" % (i, n)]
            o, _, nel = self.tslots["c%d.w" % i]
            self.w["c%d.w" % i] = self.Wb[o:o + nel].view(512, k * 512)          # master layout [n][tap][c] = GEMM layout [n][tap*c]
            if k == 3:
                self.wT["c%d.we" % i] = torch.empty(512, 1024, device=dev, dtype=bf)      # [c][ (tap 2 | tap 0) x n ]
                self.wT["c%d.wo" % i] = torch.empty(512, 512, device=dev, dtype=bf)       # tap 1
            else:
                self.wT["c%d.wt" % i] = torch.empty(k * 512, 512, device=dev, dtype=bf)   # [(tap, c)][n]
        o, _, nel = self.tslots["proj.w"]
        self.w["proj.w"] = self.Wb[o:o + nel].view(cfg.dim, 512)
        self.wT["proj.w"] = torch.empty(512, cfg.dim, device=dev, dtype=bf)
        G, cg = cfg.pos_groups, cfg.dim // cfg.pos_groups
        self.w["pos.w"] = torch.empty(G, cg, cfg.pos_k * cg, device=dev, dtype=bf)
        self.wT["pos.w"] = torch.empty(G, cg, cfg.pos_k * cg, device=dev, dtype=bf)
        self.pos_norms = torch.empty(cfg.pos_k, device=dev, dtype=torch.float32)
        self.pos_dw = torch.empty(G, cg, cfg.pos_k * cg, device=dev, dtype=torch.float32)

    def _transpose_jobs(self, tb):
        super()._transpose_jobs(tb)
        for i in range(1, 7):
            k = CONV_LAYERS[i][1]
            src = self.w["c%d.w" % i]                        # bf16 mirror of the master layout [n][tap][c]
            es = 2
            if k == 3:
                we, wo = self.wT["c%d.we" % i], self.wT["c%d.wo" % i]
                # dst[c][n] = src[n][tap][c]: a transpose of the [512 x 512] slice with row stride k*512
                tb.add(src.data_ptr() + 2 * 512 * es, we, 512, 512, ld_src=k * 512, ld_dst=1024, src_dtype=OCC_BF16_CODE)
                tb.add(src.data_ptr() + 0 * 512 * es, we.data_ptr() + 512 * 2, 512, 512, ld_src=k * 512, ld_dst=1024, src_dtype=OCC_BF16_CODE)
                tb.add(src.data_ptr() + 1 * 512 * es, wo, 512, 512, ld_src=k * 512, ld_dst=512, src_dtype=OCC_BF16_CODE)
            else:
                tb.add(src, self.wT["c%d.wt" % i], 512, k * 512, ld_src=k * 512, ld_dst=512)
        tb.add(self.w["proj.w"], self.wT["proj.w"], self.cfg.dim, 512, ld_src=512, ld_dst=self.cfg.dim)

    def refresh_operands(self, cast=True):
        super().refresh_operands(cast=cast)
        from .._lib import check, lib, ptr, stream_ptr
        cfg = self.cfg
        G, cg = cfg.pos_groups, cfg.dim // cfg.pos_groups
        sc = ops.small_scratch()
        check(lib().occ_weight_norm_pack(ptr(self.mp["pos.v"]), ptr(self.mp["pos.g"]), ptr(self.w["pos.w"]), ptr(self.wT["pos.w"]), ptr(self.pos_norms),
                                         cfg.dim, cg, cfg.pos_k, G, ptr(sc), sc.numel(), stream_ptr()), "occ_weight_norm_pack")

    def _names_extra(self):
        out = {}
        for i in range(7):
            pre = "feature_extractor.conv_layers.%d" % i
            out[pre + ".0.weight"] = ("c%d.w" % i, (lambda t, i=i: t.reshape(512, 1, 10) if i == 0 else t.permute(0, 2, 1).contiguous()))
            out[pre + ".0.bias"] = ("c%d.b" % i, None); out[pre + ".2.1.weight"] = ("c%d.g" % i, None); out[pre + ".2.1.bias"] = ("c%d.be" % i, None)
        out["layer_norm.weight"] = ("ln.g", None); out["layer_norm.bias"] = ("ln.b", None)
        out["post_extract_proj.weight"] = ("proj.w", None); out["post_extract_proj.bias"] = ("proj.b", None)
        out["encoder.pos_conv.0.weight_v"] = ("pos.v", None); out["encoder.pos_conv.0.weight_g"] = ("pos.g", lambda t: t.reshape(1, 1, -1))
        out["encoder.pos_conv.0.bias"] = ("pos.b", None)
        return out

    def ref_views(self, flat):
        out = super().ref_views(flat)
        v = lambda n: flat[self.tslots[n][0]: self.tslots[n][0] + self.tslots[n][2]].view(self.tslots[n][1])
        for i in range(7):
            pre = "feature_extractor.conv_layers.%d" % i
            out[pre + ".0.weight"] = v("c0.w").view(512, 1, 10) if i == 0 else v("c%d.w" % i).permute(0, 2, 1)       # [n][tap][c] inside
            out[pre + ".0.bias"] = v("c%d.b" % i); out[pre + ".2.1.weight"] = v("c%d.g" % i); out[pre + ".2.1.bias"] = v("c%d.be" % i)
        out["layer_norm.weight"] = v("ln.g"); out["layer_norm.bias"] = v("ln.b")
        out["post_extract_proj.weight"] = v("proj.w"); out["post_extract_proj.bias"] = v("proj.b")
        out["encoder.pos_conv.0.weight_v"] = v("pos.v"); out["encoder.pos_conv.0.weight_g"] = v("pos.g").view(1, 1, -1)
        out["encoder.pos_conv.0.bias"] = v("pos.b")
        return out

    def export_params(self):
        out = super().export_params()
        for k, (n, fn) in self._names_extra().items():
            t = self.mp[n].clone()
            out[k] = fn(t) if fn else t
        return out

    def grad_dict(self):
        out = super().grad_dict()
        for k, (n, fn) in self._names_extra().items():
            t = self.mg[n].clone()
            out[k] = fn(t) if fn else t
        return out

    # ------------------------------------------------------------------------------------------ workspaces
    def _train_ws(self, B, L):
        ws = super()._train_ws(B, L)
        tr = ws["tr"]
        if "conv" not in tr:
            cfg, dev, bf = self.cfg, self.device, torch.bfloat16
            Ts, T, M, D = ws["Ts"], ws["T"], ws["M"], cfg.dim
            e = lambda *s, dt=bf: torch.empty(*s, device=dev, dtype=dt)
            z = lambda *s, dt=bf: torch.zeros(*s, device=dev, dtype=dt)
            cv = {"act": [e(B, Ts[i], 512) for i in range(7)], "pre": [None] + [e(B * Ts[i], 512) for i in range(1, 7)],
                  "dact": [z(B, Ts[i], 512) for i in range(7)], "dpre": [None] + [z(B, Ts[i] + 2, 512) for i in range(1, 7)]}
            cv["lnfeat"] = e(M, 512)
            cv["u_pos"] = e(M, D)
            cv["dupad"] = z(B, T + cfg.pos_k, D)
            cv["dln"] = e(M, 512)
            tr["conv"] = cv
        return ws

    def _frozen_prefix(self, wav, ws, x_out=None):
        """Trainable prefix in this class: same arithmetic, but every intermediate backward needs is kept."""
        x_out = ws["x"] if x_out is None else x_out
        cfg, w = self.cfg, self.w
        cv = ws["tr"]["conv"]
        wav = wav.to(self.device, torch.float32).contiguous()
        cv["wav"] = wav
        B, L = wav.shape
        Ts, T, M, D = ws["Ts"], ws["T"], ws["M"], cfg.dim
        code = OCC_BF16_CODE
        ops.conv0_ln_gelu(wav, w["c0.w"], w["c0.b"], w["c0.g"], w["c0.be"], 10, 5, torch.bfloat16, out=cv["act"][0])
        for i in range(1, 7):
            _, k, s = CONV_LAYERS[i]
            Tin, Tout = Ts[i - 1], Ts[i]
            ops.gemm_raw(B * Tout, 512, k * 512, cv["act"][i - 1], rowmap(Tout, Tin * 512, s * 512), w["c%d.w" % i], k * 512, cv["pre"][i],
                         rowmap(B * Tout, 0, 512), code, code, bias=w["c%d.b" % i])
            ops.layernorm(cv["pre"][i], w["c%d.g" % i], w["c%d.be" % i], gelu=True, out=cv["act"][i].view(B * Tout, 512))
        ops.layernorm(cv["act"][6].view(M, 512), w["ln.g"], w["ln.b"], out=cv["lnfeat"])
        xpad = ws["xpad"]
        Tp, half = T + cfg.pos_k, cfg.pos_k // 2
        inner = xpad.data_ptr() + half * D * xpad.element_size()
        pmap = rowmap(T, Tp * D, D)
        p_in = self._p("dropout_input")
        if p_in > 0:                                             # Wav2Vec2Model.forward: features = dropout_input(post_extract_proj(features))
            tmp = cv.setdefault("proj_tmp", torch.empty(M, D, device=self.device, dtype=torch.bfloat16))
            ops.gemm_raw(M, D, 512, cv["lnfeat"], rowmap(M, 0, 512), w["proj.w"], 512, tmp, rowmap(M, 0, D), code, code, bias=w["proj.b"])
            self._drop_fwd("in", tmp, tmp, p_in)
            xpad.view(B, Tp, D)[:, half:half + T].copy_(tmp.view(B, T, D))
        else:
            ops.gemm_raw(M, D, 512, cv["lnfeat"], rowmap(M, 0, 512), w["proj.w"], 512, inner, pmap, code, code, bias=w["proj.b"])
        G, cg = cfg.pos_groups, D // cfg.pos_groups
        ops.gemm_raw(M, cg, cfg.pos_k * cg, xpad, pmap, w["pos.w"], cfg.pos_k * cg, x_out, rowmap(M, 0, D), OCC_F32, code, bias=w["pos.b"], act=ACT_GELU,
                     R=inner, r_map=pmap, r_dtype=code, a_seg=(cfg.pos_k, cg, D), groups=(G, cg, cg * cfg.pos_k * cg, cg), aux=cv["u_pos"])

    def backward(self, dfeats, grad_ready=None):
        B, L = self.ctx
        cleared = self.grads_cleared                     # (every weight gradient below is written once per backward)
        super().backward(dfeats, grad_ready=grad_ready)  # leaves d(loss)/d(encoder input) in tr["dx"]
        from .. import backend_ops as K
        from .._lib import check, lib, ptr, stream_ptr
        cfg, w = self.cfg, self.w
        ws = self._workspace(B, L, slot=0)
        tr = ws["tr"]
        cv = tr["conv"]
        Ts, T, M, D, Mp = ws["Ts"], ws["T"], ws["M"], cfg.dim, tr["Mp"]
        bfc = OCC_BF16_CODE
        G, cg, Kp = cfg.pos_groups, D // cfg.pos_groups, cfg.pos_k
        Tp = T + Kp
        dx = tr["dx"]
        xpad, dupad = ws["xpad"], cv["dupad"]
        # ---- positional conv: x = x0 + gelu(conv(x0) + b) --------------------------------------------------------
        du_in = dupad.data_ptr() + (Kp // 2 - 1) * D * 2          # interior starts 63 rows in (K/2 - 1)
        dmap = rowmap(T, Tp * D, D)
        check(lib().occ_gelu_bwd_rows(ptr(dx), ptr(cv["u_pos"]), du_in, ctypes_byref(dmap), M, D, stream_ptr()), "occ_gelu_bwd_rows")
        K.fill(self.pos_dw.view(-1), 0.0)
        # all 16 channel groups in ONE launch (grouped occ_gemm_tn: group g reads column slice g of du and of the padded input, writes
        # pos_dw[g]); the bias gradient is one column sum over the whole du
        K.gemm_tn(M, cg, Kp * cg, du_in, dmap, xpad, dmap, self.pos_dw, Kp * cg, b_seg=(Kp, cg, D), a_bf16=True, b_bf16=True, bf16_mfma=True,
                  groups=(G, cg, cg, cg * Kp * cg))
        K.colsum(du_in, dmap, M, D, self.mg["pos.b"], a_dtype=1)
        sc = ops.small_scratch()
        check(lib().occ_weight_norm_bwd(ptr(self.mp["pos.v"]), ptr(self.mp["pos.g"]), ptr(self.pos_norms), ptr(self.pos_dw), ptr(self.mg["pos.v"]),
                                        ptr(self.mg["pos.g"]), D, cg, Kp, G, ptr(sc), sc.numel(), stream_ptr()), "occ_weight_norm_bwd")
        xm = rowmap(M, 0, D)
        ops.gemm_raw(M, cg, Kp * cg, dupad, dmap, self.wT["pos.w"], Kp * cg, dx, xm, OCC_F32, bfc, R=dx, r_map=xm, r_dtype=OCC_F32,
                     a_seg=(Kp, cg, D), groups=(G, cg, cg * Kp * cg, cg))
        # ---- post_extract_proj + LayerNorm(512) --------------------------------------------------------------------
        if self._p("dropout_input") > 0:
            self._drop_bwd("in", dx, dx, self._p("dropout_input"))
        check(lib().occ_cast(ptr(dx), OCC_F32, ptr(tr["dxb"]), bfc, M * D, stream_ptr()), "occ_cast")      # dx changed since its bf16 copy was made
        self._wgrad(tr["dxb"], cv["lnfeat"], D, 512, M, "proj.w", "proj.b", c_is_zero=cleared)
        # (the bf16 copy made for the weight gradient above is also this GEMM's operand: the f32-operand kernel rounds to bf16 on its way
        # into LDS anyway, spills, and runs at a third of the eight-phase kernel's rate)
        ops.gemm_raw(M, 512, D, tr["dxb"], xm, self.wT["proj.w"], D, cv["dln"], rowmap(M, 0, 512), bfc, bfc)
        ops.layernorm_bwd_ex(cv["dln"], cv["act"][6].view(M, 512), w["ln.g"], None, None, None, cv["dact"][6].view(M, 512), None, self.mg["ln.g"], self.mg["ln.b"], gelu=False)
        # fairseq scales the gradient that enters the conv feature extractor (GradMultiply, feature_grad_mult; 0 = extractor not trained)
        fgm = self.train_cfg.feature_grad_mult
        if fgm == 0.0:
            return
        if fgm != 1.0:
            ops.dropout_ex(cv["dact"][6], cv["dact"][6], None, 0.0, scale=fgm)
        # ---- conv blocks 6..1 ------------------------------------------------------------------------------------------
        for i in range(6, 0, -1):
            _, k, s = CONV_LAYERS[i]
            Tin, Tout = Ts[i - 1], Ts[i]
            R = B * Tout
            Mpi = (R + 63) // 64 * 64
            dpre = cv["dpre"][i]
            d_in = dpre.data_ptr() + 512 * 2                                   # interior: one zero row in front of every utterance
            imap = rowmap(Tout, (Tout + 2) * 512, 512)
            ops.layernorm_bwd_ex(cv["dact"][i].view(R, 512), cv["pre"][i], w["c%d.g" % i], w["c%d.be" % i], None, None, d_in, imap,
                                 self.mg["c%d.g" % i], self.mg["c%d.be" % i], gelu=True)
            # weight gradient: dW[n][(tap,c)] = sum_m dpre[m][n] * act_{i-1}[b, s*t + tap, c]
            # (the k taps of a window are contiguous in the channels-last activation: the window IS the B row, row stride s*512)
            from .. import backend_ops as K
            K.gemm_tn(R, 512, k * 512, d_in, imap, cv["act"][i - 1], rowmap(Tout, Tin * 512, s * 512), self.mg["c%d.w" % i].view(512, k * 512), k * 512,
                      colsum_out=self.mg["c%d.b" % i], a_bf16=True, b_bf16=True, bf16_mfma=True, c_is_zero=cleared)
            # input gradient
            dprev = cv["dact"][i - 1]
            if k == 3:
                ne, no = (Tin + 1) // 2, Tin // 2
                ops.gemm_raw(B * ne, 512, 1024, dpre, rowmap(ne, (Tout + 2) * 512, 512), self.wT["c%d.we" % i], 1024, dprev, rowmap(ne, Tin * 512, 1024), bfc, bfc)
                ops.gemm_raw(B * no, 512, 512, d_in, rowmap(no, (Tout + 2) * 512, 512), self.wT["c%d.wo" % i], 512, dprev.data_ptr() + 512 * 2, rowmap(no, Tin * 512, 1024), bfc, bfc)
            else:
                ops.gemm_raw(R, k * 512, 512, d_in, imap, self.wT["c%d.wt" % i], 512, dprev, rowmap(Tout, Tin * 512, k * 512), bfc, bfc)
        # ---- conv block 0 (recomputed from the waveform) --------------------------------------------------------------------
        check(lib().occ_conv0_ln_gelu_bwd(ptr(cv["wav"]), ptr(w["c0.w"]), ptr(w["c0.b"]), ptr(w["c0.g"]), ptr(w["c0.be"]), ptr(cv["dact"][0]), bfc,
                                          ptr(self.mg["c0.w"]), ptr(self.mg["c0.b"]), ptr(self.mg["c0.g"]), ptr(self.mg["c0.be"]), B, L, Ts[0], 512, 10, 5,
                                          1e-5, stream_ptr()), "occ_conv0_ln_gelu_bwd")


def ctypes_byref(m):
    import ctypes
    return ctypes.byref(m)
