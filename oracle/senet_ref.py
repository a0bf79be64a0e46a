"""torch-CPU restatement of the SE-ResNet34 back-end (TEST ORACLE).

Follows models/senet.py:13-156.  Flat ``{name: tensor}`` parameters with the reference's
state_dict key names.
"""
import torch
import torch.nn.functional as F
from .aasist_ref import _bn

LAYERS = [3, 4, 6, 3]                   # senet.py:154-156
CHANNELS = [16, 16, 32, 64, 128]        # senet.py:66


def se_block(x, p, pre, stride, train):
    """SEBasicBlock.forward senet.py:46-61 with SELayer :23-28."""
    out = F.conv2d(x, p[pre + ".conv1.weight"], None, stride=stride, padding=1)
    out = F.relu(_bn(out, p, pre + ".bn1", train))
    out = F.conv2d(out, p[pre + ".conv2.weight"], None, padding=1)
    out = _bn(out, p, pre + ".bn2", train)
    y = out.mean(dim=(2, 3))
    y = torch.sigmoid(F.linear(F.relu(F.linear(y, p[pre + ".se.fc.0.weight"])), p[pre + ".se.fc.2.weight"]))
    out = out * y[:, :, None, None]
    if (pre + ".downsample.0.weight") in p:
        x = _bn(F.conv2d(x, p[pre + ".downsample.0.weight"], None, stride=stride), p,
                pre + ".downsample.1", train)
    return F.relu(out + x)


def senet34_forward(x, p, train=False):
    """ResNet.forward senet.py:120-142: x [B,1,T,D] -> (com[B,128], des[B,2])."""
    x = F.conv2d(x, p["conv1.weight"], None, stride=2, padding=3)
    x = F.relu(_bn(x, p, "bn1", train))
    x = F.max_pool2d(x, 3, stride=2, padding=1)
    for li, nblocks in enumerate(LAYERS):
        for bi in range(nblocks):
            stride = 2 if (li > 0 and bi == 0) else 1
            x = se_block(x, p, "layer%d.%d" % (li + 1, bi), stride, train)
    x = x.mean(dim=(2, 3))
    com = F.linear(x, p["embedding.weight"], p["embedding.bias"])
    des = F.linear(x, p["classifier.weight"], p["classifier.bias"])
    return com, des


def param_shapes():
    s = {}

    def bn(pre, c):
        s[pre + ".weight"] = (c,); s[pre + ".bias"] = (c,)
        s[pre + ".running_mean"] = (c,); s[pre + ".running_var"] = (c,)
        s[pre + ".num_batches_tracked"] = ()

    s["conv1.weight"] = (CHANNELS[0], 1, 7, 7); bn("bn1", CHANNELS[0])
    inpl = CHANNELS[0]
    for li, nblocks in enumerate(LAYERS):
        planes = CHANNELS[li + 1]
        for bi in range(nblocks):
            pre = "layer%d.%d" % (li + 1, bi)
            stride = 2 if (li > 0 and bi == 0) else 1
            s[pre + ".conv1.weight"] = (planes, inpl, 3, 3); bn(pre + ".bn1", planes)
            s[pre + ".conv2.weight"] = (planes, planes, 3, 3); bn(pre + ".bn2", planes)
            s[pre + ".se.fc.0.weight"] = (planes // 16, planes)
            s[pre + ".se.fc.2.weight"] = (planes, planes // 16)
            if stride != 1 or inpl != planes:
                s[pre + ".downsample.0.weight"] = (planes, inpl, 1, 1); bn(pre + ".downsample.1", planes)
            inpl = planes
    s["embedding.weight"] = (128, 128); s["embedding.bias"] = (128,)
    s["classifier.weight"] = (2, 128); s["classifier.bias"] = (2,)
    return s
