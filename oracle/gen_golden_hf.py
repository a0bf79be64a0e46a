#!/usr/bin/env python3
"""Writes tests/golden/xlsr_hf.npz: outputs of HuggingFace ``Wav2Vec2Model`` (oracle/hf_proxy.py) at the XLS-R-300M geometry on
seeded inputs and seeded synthetic weights -- seeds and outputs only, no weights, no source.  Run in the build container (needs
``transformers``; nothing of /root/reference is read):

    python oracle/gen_golden_hf.py

Cases: "a" = 2 transformer layers, 1 utterance of 16000 samples (every tap in full); "b" = all 24 layers, 1 utterance of 64000 samples
(final output at every 2nd frame, intermediate taps at every 4th).  tests/test_gpu_frontend.py compares the HIP f32 path with these on the
GPU box, where neither transformers nor the reference exists.  This does not lift the "parity unpinned" status of the front-end
(the reference's fairseq is absent): it is an implementation-independent cross-check."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import xlsr_ref                  # noqa: E402
from oracle.fill import fill_like            # noqa: E402
from oracle.hf_proxy import hf_model         # noqa: E402

CASES = {"a": dict(layers=2, B=1, L=16000, wseed=31, xseed=41, taps=(0, 1), stride=1),
         "b": dict(layers=24, B=1, L=64000, wseed=32, xseed=42, taps=(0, 11, 22), stride=4, out_stride=2)}


def case_inputs(c):
    cfg = xlsr_ref.XlsrConfig(dim=1024, ffn=4096, heads=16, layers=c["layers"])
    p = fill_like(xlsr_ref.param_shapes(cfg), seed=c["wseed"])
    wav = 0.1 * torch.randn(c["B"], c["L"], generator=torch.Generator().manual_seed(c["xseed"]))
    return cfg, p, wav


def main():
    out = {}
    for name, c in CASES.items():
        cfg, p, wav = case_inputs(c)
        with torch.no_grad():
            r = hf_model(cfg, p)(wav, output_hidden_states=True)
        st = c["stride"]
        out[name + "_extract_features"] = r.extract_features[:, ::st].numpy()         # conv stack output after feature_projection.layer_norm
        out[name + "_pos"] = r.hidden_states[0][:, ::st].numpy()                      # projection + positional conv (encoder input)
        for i in c["taps"]:
            if i + 1 < len(r.hidden_states) - 1:                                      # (HF returns its last hidden state already LayerNorm-ed)
                out[name + "_layer%d" % i] = r.hidden_states[i + 1][:, ::st].numpy()
        out[name + "_out"] = r.last_hidden_state[:, ::c.get("out_stride", 1)].numpy()
        out[name + "_meta"] = np.array([c["layers"], c["B"], c["L"], c["wseed"], c["xseed"], st, c.get("out_stride", 1)], dtype=np.int64)
        print(name, {k: v.shape for k, v in out.items() if k.startswith(name)})
    path = os.path.join(ROOT, "tests", "golden", "xlsr_hf.npz")
    np.savez(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
