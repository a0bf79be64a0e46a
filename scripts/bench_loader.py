"""Input-pipeline rate: PFDataset groups of 12 utterances (oc_training.py:201-256) through torch's DataLoader, host only.
Writes N synthetic 4 s / 16-bit WAV files (+ their five vocoded copies, or FLAC with --flac) to a temporary directory and measures
utterances per second for a few worker counts -- the number to hold against the GPU's consumption (bench.py: ~1150 utt/s per GPU when
fine-tuning, ~3300 frozen).  usage: python scripts/bench_loader.py [--files 48] [--flac] [--workers 0,4,8] [--rawboost 0]"""
import argparse
import os
import sys
import tempfile
import time
import wave

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=48)
    ap.add_argument("--workers", default="0,4,8")
    ap.add_argument("--flac", action="store_true", help="bona-fide / spoof files as FLAC (decoded by the library's host decoder); needs the built library")
    ap.add_argument("--rawboost", type=int, default=0, help="host-side RawBoost algo inside the Dataset (the reference's placement); 0 = off (the trainer does it on the GPU)")
    ap.add_argument("--groups", type=int, default=64)
    args = ap.parse_args()
    from occm_amd.oc_training import PFDataset, VOCODERS
    rs = np.random.RandomState(0)
    with tempfile.TemporaryDirectory() as d:
        voc = os.path.join(d, "voc"); os.makedirs(voc)
        lines = []
        for i in range(args.files):
            name = "U%04d" % i
            x = (np.clip(rs.randn(64000) * 0.1, -1, 1) * 32767).astype(np.int16)
            bona = i < args.files // 2                   # bona fide first: PFDataset indexes the protocol's first len(bonafide) lines (oc_training.py:201-212)
            targets = [os.path.join(d, name + ".wav")] + ([os.path.join(voc, "%s_%s.wav" % (v, name)) for v in VOCODERS] if bona else [])
            for path in targets:
                with wave.open(path, "wb") as w:
                    w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(x.tobytes())
            lines.append("LA_%04d %s - - %s" % (i, name, "bonafide" if bona else "spoof"))
        proto = os.path.join(d, "proto.txt")
        open(proto, "w").write("\n".join(lines) + "\n")
        ds = PFDataset(proto, d, vocoded_dir=voc, rawboost_algo=args.rawboost)
        for nw in [int(v) for v in args.workers.split(",")]:
            dl = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=True, num_workers=nw, persistent_workers=False)
            n, t0 = 0, None
            it = iter(dl)
            for g in range(args.groups):
                try:
                    x, y = next(it)
                except StopIteration:
                    it = iter(dl); x, y = next(it)
                if g == 3:
                    t0, n = time.time(), 0                   # worker start-up excluded
                n += x.shape[1]
            dt = time.time() - t0
            print("workers %2d: %8.0f utterances/s (%d groups of 12, %.2f s)" % (nw, n / dt, args.groups - 4, dt), flush=True)


if __name__ == "__main__":
    main()
