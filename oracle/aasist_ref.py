"""torch-CPU restatement of the AASIST back-end of ``AModel`` (TEST ORACLE).

Follows models/sslassist.py:58-597 (everything after ``ssl_model.extract_feat``).
Written as plain functions over a flat ``{name: tensor}`` dict that uses the
reference's state_dict key names (minus the ``ssl_model.`` subtree), so golden
state_dicts exported from the reference load without renaming.

Dropout is driven by explicit keep-masks (``masks[name]``, 0/1 tensors) so that
the HIP path and the oracle can be compared in train mode; a missing name means
"no dropout at that site".  Site names:
  GAT_layer_S, GAT_layer_T, HtrgGAT_layer_ST{11,12,21,22}   (input_drop, p=0.2)
  pool_S, pool_T, pool_hS1, pool_hT1, pool_hS2, pool_hT2    (p=0.3)
  way_T1, way_T2, way_S1, way_S2, way_M1, way_M2            (drop_way, p=0.2)
  last                                                      (p=0.5, aliases emb)
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOM = 0.1
P_GAT, P_POOL, P_WAY, P_LAST = 0.2, 0.3, 0.2, 0.5
TEMPS = {"GAT_layer_S": 2.0, "GAT_layer_T": 2.0,
         "HtrgGAT_layer_ST11": 100.0, "HtrgGAT_layer_ST12": 100.0,
         "HtrgGAT_layer_ST21": 100.0, "HtrgGAT_layer_ST22": 100.0}   # sslassist.py:441, 479-493


def _drop(x, masks, name, p):
    if masks is None or name not in masks:
        return x
    return x * masks[name].to(x.dtype) / (1.0 - p)


def _bn(x, p, pre, train, cdim=1):
    """BatchNorm over every dim but ``cdim``; train mode uses biased batch variance for the
    output and writes the unbiased one into running_var (torch semantics)."""
    w, b = p[pre + ".weight"], p[pre + ".bias"]
    shape = [1] * x.dim()
    shape[cdim] = -1
    if train:
        dims = [d for d in range(x.dim()) if d != cdim]
        mean = x.mean(dim=dims)
        var = x.var(dim=dims, unbiased=False)
        n = x.numel() // x.shape[cdim]
        with torch.no_grad():
            p[pre + ".running_mean"].mul_(1 - BN_MOM).add_(BN_MOM * mean.detach())
            p[pre + ".running_var"].mul_(1 - BN_MOM).add_(BN_MOM * var.detach() * n / max(n - 1, 1))
            if pre + ".num_batches_tracked" in p:
                p[pre + ".num_batches_tracked"] += 1
    else:
        mean, var = p[pre + ".running_mean"], p[pre + ".running_var"]
    xh = (x - mean.view(shape)) / torch.sqrt(var.view(shape) + BN_EPS)
    return xh * w.view(shape) + b.view(shape)


def _lin(x, p, pre):
    return F.linear(x, p[pre + ".weight"], p[pre + ".bias"])


def residual_block(x, p, pre, first, train):
    """Residual_block.forward, sslassist.py:406-429.  Quirk kept: bn1+selu is computed and
    thrown away (``out = self.conv1(x)`` :415); only bn1's running stats move in train mode."""
    if not first and train:
        _bn(x, p, pre + ".bn1", True)
    out = F.conv2d(x, p[pre + ".conv1.weight"], p[pre + ".conv1.bias"], padding=(1, 1))
    out = F.selu(_bn(out, p, pre + ".bn2", train))
    out = F.conv2d(out, p[pre + ".conv2.weight"], p[pre + ".conv2.bias"], padding=(0, 1))
    if (pre + ".conv_downsample.weight") in p:
        x = F.conv2d(x, p[pre + ".conv_downsample.weight"], p[pre + ".conv_downsample.bias"],
                     padding=(0, 1))
    return out + x


def _att_scores(x, p, pre, n1=None):
    """tanh(att_proj(x_i * x_j)) . att_weight  -> [B,N,N]   (sslassist.py:116-132, 272-302)."""
    pair = x.unsqueeze(2) * x.unsqueeze(1)                       # [B,N,N,D]
    a = torch.tanh(_lin(pair, p, pre + ".att_proj"))            # [B,N,N,Do]
    if n1 is None:
        return (a @ p[pre + ".att_weight"]).squeeze(-1)
    N = x.shape[1]
    type1 = (torch.arange(N) < n1)
    same1 = type1.view(N, 1) & type1.view(1, N)
    same2 = (~type1).view(N, 1) & (~type1).view(1, N)
    s11 = (a @ p[pre + ".att_weight11"]).squeeze(-1)
    s22 = (a @ p[pre + ".att_weight22"]).squeeze(-1)
    s12 = (a @ p[pre + ".att_weight12"]).squeeze(-1)
    return torch.where(same1, s11, torch.where(same2, s22, s12))


def gat_layer(x, p, pre, train, masks):
    """GraphAttentionLayer.forward, sslassist.py:84-100."""
    x = _drop(x, masks, pre, P_GAT)
    att = torch.softmax(_att_scores(x, p, pre) / TEMPS[pre], dim=-1)
    y = _lin(att @ x, p, pre + ".proj_with_att") + _lin(x, p, pre + ".proj_without_att")
    y = _bn(y, p, pre + ".bn", train, cdim=2)
    return F.selu(y)


def htrg_gat_layer(x1, x2, master, p, pre, train, masks):
    """HtrgGraphAttentionLayer.forward, sslassist.py:191-232."""
    n1, n2 = x1.shape[1], x2.shape[1]
    x = torch.cat([_lin(x1, p, pre + ".proj_type1"), _lin(x2, p, pre + ".proj_type2")], dim=1)
    if master is None:
        master = x.mean(dim=1, keepdim=True)
    x = _drop(x, masks, pre, P_GAT)
    temp = TEMPS[pre]
    att = torch.softmax(_att_scores(x, p, pre, n1) / temp, dim=-1)
    # master node: sslassist.py:255-270, 310-316
    am = torch.tanh(_lin(x * master, p, pre + ".att_projM")) @ p[pre + ".att_weightM"]   # [B,N,1]
    am = torch.softmax(am / temp, dim=-2)
    new_master = _lin(am.transpose(1, 2) @ x, p, pre + ".proj_with_attM") \
        + _lin(master, p, pre + ".proj_without_attM")
    y = _lin(att @ x, p, pre + ".proj_with_att") + _lin(x, p, pre + ".proj_without_att")
    y = F.selu(_bn(y, p, pre + ".bn", train, cdim=2))
    return y[:, :n1], y[:, n1:n1 + n2], new_master


def graph_pool(h, p, pre, k, masks, return_idx=False):
    """GraphPool.forward, sslassist.py:341-368."""
    z = _drop(h, masks, pre, P_POOL)
    scores = torch.sigmoid(_lin(z, p, pre + ".proj"))              # [B,N,1]
    n_keep = max(int(h.shape[1] * k), 1)
    _, idx = torch.topk(scores, n_keep, dim=1)
    out = torch.gather(h * scores, 1, idx.expand(-1, -1, h.shape[2]))
    return (out, idx) if return_idx else out


def backend_forward(feats, p, train=False, masks=None, taps=None):
    """AModel.forward after the SSL front-end, sslassist.py:509-597.

    feats: [B,T,1024] front-end features.  Returns (emb[B,160], logits[B,2]); in train mode with a
    ``last`` mask the returned emb is the *dropped* one (in-place alias quirk, :591-594).
    ``taps`` (optional dict) receives intermediates for layer-wise kernel tests.
    """
    x = _lin(feats, p, "LL")                                        # [B,T,128]
    x = x.transpose(1, 2).unsqueeze(1)                              # [B,1,128,T]
    x = F.max_pool2d(x, (3, 3))                                     # [B,1,42,T//3]
    x = F.selu(_bn(x, p, "first_bn", train))
    if taps is not None:
        taps["stem"] = x
    filts = [(1, 32), (32, 32), (32, 64), (64, 64), (64, 64), (64, 64)]
    for i, _ in enumerate(filts):
        x = residual_block(x, p, "encoder.%d.0" % i, i == 0, train)
        if taps is not None:
            taps["enc%d" % i] = x
    x = F.selu(_bn(x, p, "first_bn1", train))
    w = F.conv2d(x, p["attention.0.weight"], p["attention.0.bias"])
    w = _bn(F.selu(w), p, "attention.2", train)
    w = F.conv2d(w, p["attention.3.weight"], p["attention.3.bias"])
    e_S = (x * torch.softmax(w, dim=-1)).sum(dim=-1).transpose(1, 2) + p["pos_S"]   # [B,42,64]
    e_T = (x * torch.softmax(w, dim=-2)).sum(dim=-2).transpose(1, 2)                # [B,Tp,64]
    if taps is not None:
        taps["e_S"], taps["e_T"] = e_S, e_T
    gat_S = gat_layer(e_S, p, "GAT_layer_S", train, masks)
    out_S = graph_pool(gat_S, p, "pool_S", 0.5, masks)
    gat_T = gat_layer(e_T, p, "GAT_layer_T", train, masks)
    out_T = graph_pool(gat_T, p, "pool_T", 0.5, masks)
    if taps is not None:
        taps["gat_S"], taps["gat_T"], taps["out_S"], taps["out_T"] = gat_S, gat_T, out_S, out_T

    def branch(tag, master_param):
        T1, S1, m1 = htrg_gat_layer(out_T, out_S, p[master_param], p,
                                    "HtrgGAT_layer_ST%s1" % tag, train, masks)
        S1 = graph_pool(S1, p, "pool_hS%s" % tag, 0.5, masks)
        T1 = graph_pool(T1, p, "pool_hT%s" % tag, 0.5, masks)
        Ta, Sa, ma = htrg_gat_layer(T1, S1, m1, p, "HtrgGAT_layer_ST%s2" % tag, train, masks)
        return T1 + Ta, S1 + Sa, m1 + ma

    T1, S1, m1 = branch("1", "master1")
    T2, S2, m2 = branch("2", "master2")
    T1 = _drop(T1, masks, "way_T1", P_WAY); T2 = _drop(T2, masks, "way_T2", P_WAY)
    S1 = _drop(S1, masks, "way_S1", P_WAY); S2 = _drop(S2, masks, "way_S2", P_WAY)
    m1 = _drop(m1, masks, "way_M1", P_WAY); m2 = _drop(m2, masks, "way_M2", P_WAY)
    oT, oS, om = torch.max(T1, T2), torch.max(S1, S2), torch.max(m1, m2)
    emb = torch.cat([oT.abs().max(dim=1)[0], oT.mean(dim=1),
                     oS.abs().max(dim=1)[0], oS.mean(dim=1), om.squeeze(1)], dim=1)
    emb = _drop(emb, masks, "last", P_LAST)        # alias quirk: emb itself is dropped in train mode
    logits = _lin(emb, p, "out_layer")
    return emb, logits


def param_shapes():
    """Names/shapes of the 247 non-SSL state_dict entries of AModel (sslassist.py:433-504)."""
    s = {}

    def lin(pre, i, o):
        s[pre + ".weight"] = (o, i); s[pre + ".bias"] = (o,)

    def bn(pre, c):
        s[pre + ".weight"] = (c,); s[pre + ".bias"] = (c,)
        s[pre + ".running_mean"] = (c,); s[pre + ".running_var"] = (c,)
        s[pre + ".num_batches_tracked"] = ()

    def conv(pre, ci, co, kh, kw):
        s[pre + ".weight"] = (co, ci, kh, kw); s[pre + ".bias"] = (co,)

    lin("LL", 1024, 128)
    bn("first_bn", 1); bn("first_bn1", 64)
    for i, (ci, co) in enumerate([(1, 32), (32, 32), (32, 64), (64, 64), (64, 64), (64, 64)]):
        pre = "encoder.%d.0" % i
        if i > 0:
            bn(pre + ".bn1", ci)
        conv(pre + ".conv1", ci, co, 2, 3)
        bn(pre + ".bn2", co)
        conv(pre + ".conv2", co, co, 2, 3)
        if ci != co:
            conv(pre + ".conv_downsample", ci, co, 1, 3)
    conv("attention.0", 64, 128, 1, 1); bn("attention.2", 128); conv("attention.3", 128, 64, 1, 1)
    s["pos_S"] = (1, 42, 64); s["master1"] = (1, 1, 64); s["master2"] = (1, 1, 64)
    for pre in ("GAT_layer_S", "GAT_layer_T"):
        lin(pre + ".att_proj", 64, 64); s[pre + ".att_weight"] = (64, 1)
        lin(pre + ".proj_with_att", 64, 64); lin(pre + ".proj_without_att", 64, 64)
        bn(pre + ".bn", 64)
    for tag, di in (("ST11", 64), ("ST12", 32), ("ST21", 64), ("ST22", 32)):
        pre = "HtrgGAT_layer_" + tag
        lin(pre + ".proj_type1", di, di); lin(pre + ".proj_type2", di, di)
        lin(pre + ".att_proj", di, 32); lin(pre + ".att_projM", di, 32)
        for w in ("att_weight11", "att_weight22", "att_weight12", "att_weightM"):
            s[pre + "." + w] = (32, 1)
        lin(pre + ".proj_with_att", di, 32); lin(pre + ".proj_without_att", di, 32)
        lin(pre + ".proj_with_attM", di, 32); lin(pre + ".proj_without_attM", di, 32)
        bn(pre + ".bn", 32)
    lin("pool_S.proj", 64, 1); lin("pool_T.proj", 64, 1)
    for n in ("pool_hS1", "pool_hT1", "pool_hS2", "pool_hT2"):
        lin(n + ".proj", 32, 1)
    lin("out_layer", 160, 2)
    return s
