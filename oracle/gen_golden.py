#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own Python in the build container.

Usage:  python oracle/gen_golden.py [/root/reference]

The reference cannot travel to the GPU box, so only inputs-by-seed and expected outputs are
stored (data, no reference source).  Modules are loaded by file path with empty stand-in
modules for imports the image lacks (fairseq, librosa: the pieces exercised here never call
them); ``sslassist.SSLModel`` is replaced by a parameter-free stub so ``AModel`` can be built
without a fairseq checkpoint -- the front-end is therefore NOT covered by these fixtures
(see oracle/xlsr_ref.py: parity unpinned).
"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import aasist_ref, lcnn_ref, senet_ref  # noqa: E402  (shape tables only)
from oracle.fill import fill_like                    # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
warnings.filterwarnings("ignore", category=DeprecationWarning)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _stub(name):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
    return sys.modules[name]


def synth_wave(seed, n):
    return (np.random.RandomState(seed).randn(n) * 0.1).astype(np.float32)


# ---------------------------------------------------------------------------------------------
def gen_rawboost():
    sys.path.insert(0, REF)
    import RawBoost as RB
    _stub("librosa")
    du = _load("ref_data_utils_SSL", os.path.join(REF, "data_utils_SSL.py"))
    from oracle.rawboost_np import RawBoostArgs
    args = RawBoostArgs()
    out = {}
    L = 8000
    for algo in range(0, 9):
        for seed in (11, 12):
            x = synth_wave(1000 + seed, L)
            if algo == 3:
                x = x * 12.0 if seed == 12 else x          # exercise peak>1 handling downstream
            np.random.seed(seed)
            y = du.process_Rawboost_feature(x, 16000, args, algo)
            out["algo%d_seed%d" % (algo, seed)] = np.asarray(y)
    # a loud input so the LnL/ISD "peak > 1" normalisation branch is taken
    x = synth_wave(77, L) * 8.0
    np.random.seed(5)
    out["algo5_loud"] = np.asarray(du.process_Rawboost_feature(x, 16000, args, 5))
    # filter design alone
    for seed in (1, 2, 3):
        np.random.seed(seed)
        out["notch_seed%d" % seed] = RB.genNotchCoeffs(5, 20, 8000, 100, 1000, 10, 100, 0, 0, 16000)
    np.random.seed(4)
    out["notch_gain_seed4"] = RB.genNotchCoeffs(5, 20, 8000, 100, 1000, 10, 100, -5, -20, 16000)
    # parameter-injected FIR
    rs = np.random.RandomState(9)
    for nt in (11, 101, 501):
        b = rs.randn(nt) / nt
        out["fir_%d" % nt] = RB.filterFIR(synth_wave(90 + nt, L), b)
    # pad()
    out["pad_short"] = du.pad(synth_wave(3, 1000), 2600)
    out["pad_long"] = du.pad(synth_wave(3, 3000), 2600)
    np.savez_compressed(os.path.join(OUT, "rawboost.npz"), **out)
    print("rawboost.npz", len(out))


def gen_losses_eer():
    cl = _load("ref_custom_loss", os.path.join(REF, "losses", "custom_loss.py"))
    em = _load("ref_evaluate_metrics", os.path.join(REF, "evaluate_metrics.py"))
    out = {}
    for seed, (n, e) in enumerate([(12, 160), (12, 128), (6, 160), (24, 32)]):
        g = torch.Generator().manual_seed(seed)
        emb = torch.randn(n, e, generator=g)
        logits = torch.randn(n, 2, generator=g)
        labels = (torch.arange(n) % 12 >= 6).long()
        out["compact_%d" % seed] = cl.compactness_loss(emb).numpy()
        out["descr_%d" % seed] = cl.descriptiveness_loss(logits, labels).numpy()
        out["triplet_%d" % seed] = cl.triplet_loss(emb).numpy()
        out["euclid_%d" % seed] = cl.euclidean_distance_loss(emb).numpy()
        # gradients of 0.1*c + 0.9*d (test_dataloader_v2.py:127 weighting)
        emb.requires_grad_(True); logits.requires_grad_(True)
        (0.1 * cl.compactness_loss(emb) + 0.9 * cl.descriptiveness_loss(logits, labels)).backward()
        out["gemb_%d" % seed] = emb.grad.numpy()
        out["glogits_%d" % seed] = logits.grad.numpy()
    for seed in (0, 1, 2):
        rs = np.random.RandomState(seed)
        tar = rs.randn(700) + 1.0
        non = rs.randn(1300) - 0.5
        if seed == 2:                                   # ties
            tar = np.round(tar, 1); non = np.round(non, 1)
        eer, thr = em.compute_eer(tar, non)
        out["eer_%d" % seed] = np.array([eer, thr])
        out["conf_%d" % seed] = np.array(em.calculate_confusion_matrix(tar, non, thr))
    np.savez_compressed(os.path.join(OUT, "losses_eer.npz"), **out)
    print("losses_eer.npz", len(out))


def _build_ref_amodel():
    _stub("fairseq")
    sa = _load("ref_sslassist", os.path.join(REF, "models", "sslassist.py"))

    class _NoSSL(torch.nn.Module):                      # parameter-free stand-in for the fairseq wrapper
        def __init__(self, device):
            super().__init__()
            self.out_dim = 1024

        def extract_feat(self, x):
            return x                                    # features are fed in directly

    sa.SSLModel = _NoSSL
    m = sa.AModel(None, "cpu")
    return m


def gen_aasist():
    cl = sys.modules.get("ref_custom_loss") or _load("ref_custom_loss", os.path.join(REF, "losses", "custom_loss.py"))
    m = _build_ref_amodel()
    params = fill_like(aasist_ref.param_shapes(), seed=0)
    m.load_state_dict(params, strict=True)              # proves the key/shape table equals the reference's
    out = {}
    for tag, (B, T, fseed) in {"a": (12, 199, 100), "b": (3, 201, 101), "c": (1, 650, 102)}.items():
        g = torch.Generator().manual_seed(fseed)
        feats = torch.randn(B, T, 1024, generator=g)
        m.load_state_dict(params, strict=True)
        m.eval()
        with torch.no_grad():
            emb, logit = m(feats)
        out["eval_emb_" + tag] = emb.numpy(); out["eval_out_" + tag] = logit.numpy()
        if B < 2:
            continue
        # train mode, dropout forced to p=0 (module attribute, not a source change)
        m.train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        m.zero_grad()
        emb, logit = m(feats)
        labels = (torch.arange(B) % 12 >= 6).long()
        loss = 0.1 * cl.compactness_loss(emb) + 0.9 * cl.descriptiveness_loss(logit, labels)
        loss.backward()
        out["train_emb_" + tag] = emb.detach().numpy(); out["train_out_" + tag] = logit.detach().numpy()
        out["train_loss_" + tag] = loss.detach().numpy()
        sd = m.state_dict()
        for k in ("first_bn.running_mean", "first_bn.running_var", "encoder.1.0.bn1.running_mean",
                  "encoder.1.0.bn1.running_var", "encoder.3.0.bn2.running_var", "first_bn1.running_mean",
                  "attention.2.running_var", "GAT_layer_T.bn.running_mean", "HtrgGAT_layer_ST12.bn.running_var"):
            out["rs_%s_%s" % (tag, k)] = sd[k].numpy().copy()
        names, norms = [], []
        for k, v in m.named_parameters():
            names.append(k); norms.append(float(v.grad.norm()) if v.grad is not None else -1.0)
        out["gradnames_" + tag] = np.array(names); out["gradnorms_" + tag] = np.array(norms)
        for k in ("out_layer.weight", "pos_S", "master1", "encoder.0.0.conv1.weight", "GAT_layer_S.att_weight",
                  "HtrgGAT_layer_ST11.att_weight12", "pool_T.proj.weight", "attention.3.bias", "first_bn.weight",
                  "encoder.2.0.conv_downsample.weight", "HtrgGAT_layer_ST22.proj_with_attM.weight"):
            out["grad_%s_%s" % (tag, k)] = dict(m.named_parameters())[k].grad.numpy().copy()
        out["grad_%s_LL.weight_rows0_4" % tag] = m.LL.weight.grad[:4].numpy().copy()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.2                              # value irrelevant in eval; restored per use below
    np.savez_compressed(os.path.join(OUT, "aasist.npz"), **out)
    print("aasist.npz", len(out))


def gen_train_steps():
    """Three optimizer steps of the reference's own loop body (oc_training.py:363-385: zero_grad -> forward -> 0.1 c + 0.9 d ->
    backward -> Adam) on fixed features, dropout p forced to 0: per-step losses and a few parameters / buffers afterwards."""
    cl = sys.modules.get("ref_custom_loss") or _load("ref_custom_loss", os.path.join(REF, "losses", "custom_loss.py"))
    m = _build_ref_amodel()
    m.load_state_dict(fill_like(aasist_ref.param_shapes(), seed=0), strict=True)
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    labels = (torch.arange(12) >= 6).long()
    out = {"loss_c": [], "loss_d": []}
    for step in range(3):
        feats = torch.randn(12, 199, 1024, generator=torch.Generator().manual_seed(200 + step))
        opt.zero_grad()
        emb, logit = m(feats)
        lc, ld = cl.compactness_loss(emb), cl.descriptiveness_loss(logit, labels)
        (0.1 * lc + 0.9 * ld).backward()
        opt.step()
        out["loss_c"].append(float(lc)); out["loss_d"].append(float(ld))
    sd = m.state_dict()
    for k in ("out_layer.weight", "out_layer.bias", "LL.bias", "pos_S", "master1", "first_bn.running_mean", "first_bn.num_batches_tracked",
              "encoder.5.0.conv2.bias", "GAT_layer_T.att_weight"):
        out["p_" + k] = sd[k].numpy().copy()
    out["loss_c"] = np.array(out["loss_c"]); out["loss_d"] = np.array(out["loss_d"])
    np.savez_compressed(os.path.join(OUT, "train_steps.npz"), **out)
    print("train_steps.npz", out["loss_c"], out["loss_d"])


def gen_senet():
    _stub("fairseq")
    sys.path.insert(0, os.path.join(REF, "models"))
    xl = types.ModuleType("xlsr"); xl.SSLModel = object; sys.modules["xlsr"] = xl
    se = _load("ref_senet", os.path.join(REF, "models", "senet.py"))
    m = se.se_resnet34()
    params = fill_like(senet_ref.param_shapes(), seed=1)
    m.load_state_dict(params, strict=True)
    out = {}
    for tag, shp, s in (("lfcc", (4, 1, 266, 13), 7), ("ssl", (2, 1, 199, 1024), 8)):
        g = torch.Generator().manual_seed(s)
        x = torch.randn(*shp, generator=g)
        m.load_state_dict(params, strict=True)
        m.eval()
        with torch.no_grad():
            com, des = m(x)
        out["eval_com_" + tag] = com.numpy(); out["eval_des_" + tag] = des.numpy()
        m.train()
        com, des = m(x)
        out["train_com_" + tag] = com.detach().numpy(); out["train_des_" + tag] = des.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "senet.npz"), **out)
    print("senet.npz", len(out))


def gen_lcnn():
    """models/lcnn.py ``lcnn_net(asoftmax=False)`` (the form occm.py:52 and lcnn.py:244 build): eval outputs, and a train-mode pass with
    the Dropout probabilities forced to 0 (module attribute, not a source change): outputs, BatchNorm running statistics afterwards, every
    parameter-gradient norm and three whole gradients of sum(logits * fixed weights)."""
    _stub("fairseq")
    sys.path.insert(0, os.path.join(REF, "models"))
    xl = types.ModuleType("xlsr"); xl.SSLModel = object; sys.modules["xlsr"] = xl
    lc = _load("ref_lcnn", os.path.join(REF, "models", "lcnn.py"))
    m = lc.lcnn_net(asoftmax=False)
    params = fill_like(lcnn_ref.param_shapes(), seed=4)
    m.load_state_dict(params, strict=True)              # proves the key/shape table equals the reference's
    out = {}
    for tag, shp, s in (("a", (3, 1, 48, 1024), 21), ("b", (2, 1, 199, 1024), 22)):
        g = torch.Generator().manual_seed(s)
        x = torch.randn(*shp, generator=g)
        m.load_state_dict(params, strict=True)
        m.eval()
        with torch.no_grad():
            out["eval_" + tag] = m(x).numpy()
        m.train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        m.zero_grad()
        y = m(x)
        wgt = torch.randn(y.shape, generator=g)
        (y * wgt).sum().backward()
        out["train_" + tag] = y.detach().numpy(); out["train_wgt_" + tag] = wgt.numpy()
        sd = m.state_dict()
        for k in ("layer2.2.running_mean", "layer2.2.running_var", "layer3.2.running_mean", "layer3.2.running_var"):
            out["rs_%s_%s" % (tag, k)] = sd[k].numpy().copy()
        names, norms = [], []
        for k, v in m.named_parameters():
            names.append(k); norms.append(float(v.grad.norm()) if v.grad is not None else -1.0)
        out["gradnames_" + tag] = np.array(names); out["gradnorms_" + tag] = np.array(norms)
        for k in ("layer1.0.filter.weight", "layer3.0.conv.filter.weight", "fc0.0.filter.0.weight"):
            out["grad_%s_%s" % (tag, k)] = dict(m.named_parameters())[k].grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "lcnn.npz"), **out)
    print("lcnn.npz", len(out))


def gen_protocol():
    _stub("librosa")
    du = sys.modules.get("ref_data_utils_SSL") or _load("ref_data_utils_SSL", os.path.join(REF, "data_utils_SSL.py"))
    lines = ["LA_0079 LA_T_1138215 - - bonafide", "LA_0079 LA_T_1271820 - - bonafide",
             "LA_0081 LA_T_5367204 - A01 spoof", "LA_0082 LA_T_9557645 - A04 spoof",
             "LA_0083 LA_T_2217429 - - bonafide"]
    path = os.path.join(OUT, "protocol_train.txt")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    d, l = du.genSpoof_list(path, is_train=True)
    out = {"keys": np.array(l), "labels": np.array([d[k] for k in l])}
    np.savez_compressed(os.path.join(OUT, "protocol.npz"), **out)
    print("protocol.npz")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gen_rawboost()
    gen_losses_eer()
    gen_aasist()
    gen_train_steps()
    gen_senet()
    gen_lcnn()
    gen_protocol()
