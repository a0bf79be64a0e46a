"""Deterministic parameter filler shared by the golden generator, the tests and bench.py.

No checkpoint exists for this path, so every weight is synthetic:
tensors are visited in sorted-name order and drawn from one seeded ``torch.Generator``.
"""
import math
import torch


def fill_like(shapes, seed=0, dtype=torch.float32):
    """shapes: {name: tuple}.  Returns {name: tensor} with sane magnitudes per tensor kind."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[name] = torch.zeros((), dtype=torch.int64)
            continue
        r = torch.randn(shp, generator=g, dtype=torch.float32)
        if leaf == "running_var":
            t = 0.5 + r.abs()
        elif leaf == "running_mean":
            t = 0.1 * r
        elif leaf == "weight_g":
            t = 1.0 + 0.1 * r
        elif len(shp) <= 1 and leaf == "weight":           # norm scales
            t = 1.0 + 0.1 * r
        elif len(shp) <= 1:                                # biases
            t = 0.05 * r
        elif leaf in ("pos_S", "master1", "master2"):
            t = r
        else:                                              # linear / conv kernels: ~1/sqrt(fan_in)
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            if leaf.startswith("att_weight"):
                fan_in = shp[0]
            t = r / math.sqrt(max(fan_in, 1))
        out[name] = t.to(dtype)
    return out
