"""torch-CPU restatement of the LCNN back-end and the dual-branch OCCM head (TEST ORACLE).

Follows models/lcnn.py:121-241 (mfm :121-136, group :139-150, LCNN :152-217 with ``asoftmax=False``, the only form the reference
instantiates: lcnn.py:244, occm.py:52) and models/occm.py:48-67.  Flat ``{name: tensor}`` parameters with the reference's
state_dict key names.  Pinned by tests/golden/lcnn.npz, which oracle/gen_golden.py produced by running the reference's own
``lcnn_net(asoftmax=False)`` (tests/test_oracle_backends.py).
"""
import torch
import torch.nn.functional as F
from .aasist_ref import _bn

C_S = [128, 64, 32, 16, 8, 4, 2]        # lcnn.py:153
P_FC = (0.75, 0.75, 0.0)                # Dropout of fc0 / fc1 / fc2 (lcnn.py:172-181)


def _drop(x, masks, site, p):
    if masks is None or p <= 0 or site not in masks:
        return x
    return x * masks[site].to(x.dtype).reshape(x.shape) / (1.0 - p)


def mfm_conv(x, p, pre, k, pad):
    """mfm(type=1): Conv2d(in, 2*out) then the max of the two channel halves (lcnn.py:126-136)."""
    y = F.conv2d(x, p[pre + ".filter.weight"], p[pre + ".filter.bias"], padding=pad)
    a, b = torch.split(y, y.shape[1] // 2, 1)
    return torch.max(a, b)


def mfm_fc(x, p, pre, masks, site, pdrop, train):
    """mfm(type=0): Linear(in, 2*out) -> Dropout -> max of the two halves."""
    y = F.linear(x, p[pre + ".filter.0.weight"], p[pre + ".filter.0.bias"])
    if train:
        y = _drop(y, masks, site, pdrop)
    a, b = torch.split(y, y.shape[1] // 2, 1)
    return torch.max(a, b)


def group(x, p, pre):
    """group.forward (lcnn.py:147-150): conv_a (1x1 mfm) -> conv (3x3 mfm); its ``bn`` member is never applied."""
    x = mfm_conv(x, p, pre + ".conv_a", 1, 0)
    return mfm_conv(x, p, pre + ".conv", 3, 1)


def lcnn_forward(x, p, train=False, masks=None, taps=None):
    """LCNN.forward (lcnn.py:186-214): x [B,1,T,D] -> logits [B,2].  Train mode: batch statistics in the two BatchNorm2d and the
    dropouts of fc0/fc1 through explicit keep-masks ``masks`` {"fc0", "fc1"} (absent = no dropout at that site)."""
    x = F.max_pool2d(mfm_conv(x, p, "layer1.0", 5, 2), 2, 2)
    if taps is not None:
        taps["l1"] = x
    x = _bn(F.max_pool2d(group(x, p, "layer2.0"), 2, 2), p, "layer2.2", train)
    if taps is not None:
        taps["l2"] = x
    x = _bn(F.max_pool2d(group(x, p, "layer3.0"), 2, 2), p, "layer3.2", train)
    if taps is not None:
        taps["l3"] = x
    x = F.adaptive_avg_pool2d(x, (1, 64)).reshape(x.shape[0], -1)
    x = mfm_fc(x, p, "fc0.0", masks, "fc0", P_FC[0], train)
    x = mfm_fc(x, p, "fc1.0", masks, "fc1", P_FC[1], train)
    x = mfm_fc(x, p, "fc2.0", masks, "fc2", P_FC[2], train)
    return F.linear(x, p["fc3.weight"], p["fc3.bias"])


def param_shapes():
    s = {}

    def conv(pre, ci, co, k):
        s[pre + ".filter.weight"] = (2 * co, ci, k, k); s[pre + ".filter.bias"] = (2 * co,)

    def bn(pre, c):
        s[pre + ".weight"] = (c,); s[pre + ".bias"] = (c,)
        s[pre + ".running_mean"] = (c,); s[pre + ".running_var"] = (c,)
        s[pre + ".num_batches_tracked"] = ()

    def grp(pre, ci, co):
        conv(pre + ".conv_a", ci, ci, 1); bn(pre + ".bn", ci); conv(pre + ".conv", ci, co, 3)

    def fc(pre, i, o):
        s[pre + ".filter.0.weight"] = (2 * o, i); s[pre + ".filter.0.bias"] = (2 * o,)

    conv("layer1.0", 1, C_S[5], 5)
    grp("layer2.0", C_S[5], C_S[4]); bn("layer2.2", C_S[4])
    grp("layer3.0", C_S[4], C_S[3]); bn("layer3.2", C_S[3])
    fc("fc0.0", 1024, 32); fc("fc1.0", 32, 32); fc("fc2.0", 32, 8)
    s["fc3.weight"] = (2, 8); s["fc3.bias"] = (2,)
    return s


def occm_forward(feats, p_senet, p_lcnn, train=False, masks=None):
    """OCCM.forward after the front-end (occm.py:55-67): both branches see x.unsqueeze(1); returns (senet34 (com, des), lcnn logits)."""
    from .senet_ref import senet34_forward
    x = feats.unsqueeze(1)
    return senet34_forward(x, p_senet, train), lcnn_forward(x, p_lcnn, train, masks)
