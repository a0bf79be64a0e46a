"""torch-CPU restatement of the wav2vec 2.0 / XLS-R feature path (TEST ORACLE).

**Parity unpinned vs the reference**: the arithmetic behind
``SSLModel.extract_feat`` (models/sslassist.py:31-49, models/xlsr.py:39-48) lives in
third-party fairseq @ a54021305d6b3c4c5959ac9395135f63202db8f1 (requirement.txt:33), which is
neither vendored in the reference nor installed here, and the reference holds no test vector for
it.  This file restates the published algorithm of
``fairseq.models.wav2vec.wav2vec2.Wav2Vec2Model.forward(source, mask=False, features_only=True)``
for the XLS-R configuration (extractor_mode="layer_norm", conv_bias=True,
layer_norm_first=True, conv_pos=128, conv_pos_groups=16) and is pinned only against
HuggingFace ``transformers.Wav2Vec2Model`` (tests/test_oracle_xlsr.py).

Parameters: flat ``{name: tensor}`` with fairseq's state_dict names (the names found under
``ssl_model.model.`` in the reference's checkpoints).
"""
import math
import torch
import torch.nn.functional as F

CONV_LAYERS = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2


class XlsrConfig:
    def __init__(self, dim=1024, ffn=4096, heads=16, layers=24, conv_dim=512,
                 pos_k=128, pos_groups=16):
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


def pos_conv_weight(p):
    """torch weight_norm with dim=2: w = g * v / ||v||, norm taken over dims (0,1) per kernel tap."""
    v = p["encoder.pos_conv.0.weight_v"]
    g = p["encoder.pos_conv.0.weight_g"]
    return g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()


def conv_features(wav, p, upto=None):
    """ConvFeatureExtractionModel (mode layer_norm): conv -> LayerNorm over channels -> GELU."""
    x = wav.unsqueeze(1)                                           # [B,1,L]
    for i, (c, k, s) in enumerate(CONV_LAYERS):
        pre = "feature_extractor.conv_layers.%d" % i
        x = F.conv1d(x, p[pre + ".0.weight"], p[pre + ".0.bias"], stride=s)
        x = F.layer_norm(x.transpose(1, 2), (c,), p[pre + ".2.1.weight"], p[pre + ".2.1.bias"]).transpose(1, 2)
        x = F.gelu(x)
        if upto is not None and i == upto:
            break
    return x.transpose(1, 2)                                       # [B,T,512]


def _drop(x, masks, site, p):
    """F.dropout with an explicit keep-mask (u8, x's shape): x * keep / (1 - p); site absent or p == 0 = identity."""
    if masks is None or p <= 0 or site not in masks:
        return x
    return x * masks[site].to(x.dtype).reshape(x.shape) / (1.0 - p)


def encoder_layer(x, p, pre, heads, masks=None, site=None, p_res=0.0, p_act=0.0, p_att=0.0):
    """TransformerSentenceEncoderLayer with layer_norm_first=True (pre-LN).  Train mode (masks given): dropout1 on the attention
    branch, dropout2 (activation_dropout) after the activation, dropout3 on the FFN branch -- fairseq wav2vec2.py
    TransformerSentenceEncoderLayer.forward."""
    B, T, D = x.shape
    hd = D // heads
    h = F.layer_norm(x, (D,), p[pre + ".self_attn_layer_norm.weight"], p[pre + ".self_attn_layer_norm.bias"])
    q = F.linear(h, p[pre + ".self_attn.q_proj.weight"], p[pre + ".self_attn.q_proj.bias"]) * hd ** -0.5
    k = F.linear(h, p[pre + ".self_attn.k_proj.weight"], p[pre + ".self_attn.k_proj.bias"])
    v = F.linear(h, p[pre + ".self_attn.v_proj.weight"], p[pre + ".self_attn.v_proj.bias"])
    q = q.view(B, T, heads, hd).transpose(1, 2)
    k = k.view(B, T, heads, hd).transpose(1, 2)
    v = v.view(B, T, heads, hd).transpose(1, 2)
    # MultiheadAttention: attn_probs = dropout(softmax(q k^T), p = attention_dropout) -- keep-mask site "l<i>.att" [B,H,T,T]
    a = _drop(torch.softmax(q @ k.transpose(-1, -2), dim=-1), masks, "%s.att" % site, p_att) @ v         # [B,H,T,hd]
    a = a.transpose(1, 2).reshape(B, T, D)
    y = F.linear(a, p[pre + ".self_attn.out_proj.weight"], p[pre + ".self_attn.out_proj.bias"])
    x = x + _drop(y, masks, "%s.d1" % site, p_res)
    h = F.layer_norm(x, (D,), p[pre + ".final_layer_norm.weight"], p[pre + ".final_layer_norm.bias"])
    h = _drop(F.gelu(F.linear(h, p[pre + ".fc1.weight"], p[pre + ".fc1.bias"])), masks, "%s.act" % site, p_act)
    return x + _drop(F.linear(h, p[pre + ".fc2.weight"], p[pre + ".fc2.bias"]), masks, "%s.d3" % site, p_res)


def extract_feat(wav, p, cfg, taps=None, train=None):
    """wav [B,L] (un-normalised, as the reference feeds it) -> [B,T,dim].

    train (optional): fairseq's train-mode behaviour, active in the reference because ``aasist.train()`` (oc_training.py:351) also
    puts the never-eval()-ed SSLModel (sslassist.py:20-29) in train mode.  A dict with the probabilities ``dropout``,
    ``activation_dropout``, ``attention_dropout``, ``dropout_input``, ``feature_grad_mult``, the explicit keep-masks ``masks`` {site: u8}
    (sites "in", "enc", "l<i>.d1", "l<i>.act", "l<i>.d3", "l<i>.att" [B,H,T,T]) and the layerdrop decisions ``keep`` [bool per layer] -- Wav2Vec2Model.forward and
    TransformerEncoder.extract_features of fairseq @ a540213."""
    tr = train or {}
    masks, keep = tr.get("masks"), tr.get("keep")
    f = conv_features(wav, p)
    if taps is not None:
        taps["conv"] = f
    fgm = tr.get("feature_grad_mult", 1.0)
    if fgm == 0.0:                                                 # fairseq runs the extractor under no_grad
        f = f.detach()
    elif fgm != 1.0:                                               # GradMultiply.apply(features, fgm): identity forward, gradient x fgm
        f = f.detach() * (1.0 - fgm) + f * fgm
    f = F.layer_norm(f, (cfg.conv_dim,), p["layer_norm.weight"], p["layer_norm.bias"])
    x = F.linear(f, p["post_extract_proj.weight"], p["post_extract_proj.bias"])
    x = _drop(x, masks, "in", tr.get("dropout_input", 0.0))
    if taps is not None:
        taps["proj"] = x
    pc = F.conv1d(x.transpose(1, 2), pos_conv_weight(p), p["encoder.pos_conv.0.bias"],
                  padding=cfg.pos_k // 2, groups=cfg.pos_groups)
    if cfg.pos_k % 2 == 0:
        pc = pc[:, :, :-1]                                         # SamePad
    x = x + F.gelu(pc).transpose(1, 2)
    x = _drop(x, masks, "enc", tr.get("dropout", 0.0))
    if taps is not None:
        taps["pos"] = x
    for i in range(cfg.layers):
        if keep is None or keep[i]:
            x = encoder_layer(x, p, "encoder.layers.%d" % i, cfg.heads, masks, "l%d" % i, tr.get("dropout", 0.0), tr.get("activation_dropout", 0.0),
                              tr.get("attention_dropout", 0.0))
        if taps is not None:
            taps["layer%d" % i] = x
    return F.layer_norm(x, (cfg.dim,), p["encoder.layer_norm.weight"], p["encoder.layer_norm.bias"])


def flops_forward(L, cfg):
    """Algorithmic forward FLOPs per utterance (SURVEY.md section 8d formula)."""
    Ts, Lc = [], L
    for _, k, s in CONV_LAYERS:
        Lc = (Lc - k) // s + 1
        Ts.append(Lc)
    C, T, d, f, n = cfg.conv_dim, Ts[-1], cfg.dim, cfg.ffn, cfg.layers
    fe = 2 * (Ts[0] * C * 10 + sum(Ts[1:5]) * C * C * 3 + sum(Ts[5:7]) * C * C * 2)
    proj = 2 * T * C * d
    pos = 2 * (T + 1) * d * (d // cfg.pos_groups) * cfg.pos_k
    lin = 2 * T * (4 * d * d + 2 * d * f) * n
    att = 4 * T * T * d * n
    return {"fe": fe, "proj": proj, "pos": pos, "lin": lin, "att": att,
            "total": fe + proj + pos + lin + att}
