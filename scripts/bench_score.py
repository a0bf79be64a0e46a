"""Scoring throughput (oc_classifier.py:159-202, 243-265 in the reference): utterances/s of embed_dataset -- the work
of create_reference_embedding2 / score_eval_set_1c2 -- on synthetic trials of MIXED length (uniform 1.0-9.0 s, mean 5 s; ASVspoof LA eval
files run 0.5-13 s), XLS-R-300M x 24 layers + AASIST, waveforms resident in host memory (no file decoding in the timed region).

    python scripts/bench_score.py [--n 96] [--dtypes f32,bf16] [--batches 1,8,16]

One JSON line per (dtype, batch): utterances/s, audio seconds per second, the front-end's algorithmic FLOP rate against the MFMA peak of the
arithmetic in use (exact-f32 MFMA 157 TFLOP/s, bf16 2500), padding overhead of the masked batches."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from occm_amd.models import xlsr
from occm_amd.models.sslassist import AModel
from occm_amd.oc_classifier import embed_dataset, n_frames

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=96)
ap.add_argument("--dtypes", default="f32,f32x3,bf16")
ap.add_argument("--batches", default="1,8,16")
ap.add_argument("--equal", action="store_true", help="also time the un-masked fallback (batches of equal frame count only)")
args = ap.parse_args()


class Trials(torch.utils.data.Dataset):
    def __init__(self, n, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.lens = [int(v) for v in torch.randint(16000, 144000, (n,), generator=g)]
        self.wavs = [0.1 * torch.randn(L, generator=g) for L in self.lens]

    def __len__(self):
        return len(self.wavs)

    def __getitem__(self, i):
        return self.wavs[i], torch.zeros(1, dtype=torch.int64)


def fe_flops(L, cfg):
    T = n_frames(L)
    Ts, Lc = [], L
    for _, k, s in xlsr.CONV_LAYERS:
        Lc = (Lc - k) // s + 1; Ts.append(Lc)
    C, d, f, nl = 512, cfg.dim, cfg.ffn, cfg.layers
    fe = 2 * (Ts[0] * C * 10 + sum(Ts[1:5]) * C * C * 3 + (Ts[5] + Ts[6]) * C * C * 2)
    return fe + 2 * T * C * d + 2 * (T + 1) * d * (d // 16) * 128 + 2 * T * (4 * d * d + 2 * d * f) * nl + 4 * T * T * d * nl


ds = Trials(args.n)
loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False)
cfg = xlsr.XlsrConfig.xlsr_300m()
flops = sum(fe_flops(L, cfg) for L in ds.lens)
audio_s = sum(ds.lens) / 16000.0
for dname in args.dtypes.split(","):
    dt = {"f32": torch.float32, "f32x3": torch.float32, "bf16": torch.bfloat16}[dname]
    model = AModel(None, "cuda", ssl_cfg=cfg, ssl_dtype=dt, synthetic_ssl=True, ssl_f32_gemm="x3" if dname == "f32x3" else "exact")
    model.eval()
    base = None
    for bs in [int(v) for v in args.batches.split(",")]:
        for masked in ([True, False] if (args.equal and bs > 1) else [True]):
            embed_dataset(model, loader, "cuda", bs, masked=masked)      # warm-up pass over the same trials: a long evaluation runs with every workspace shape already allocated
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            emb, _ = embed_dataset(model, loader, "cuda", bs, masked=masked)
            torch.cuda.synchronize()
            dtm = time.perf_counter() - t0
            if base is None:
                base = emb.clone()
            peak = 157.3 if dname == "f32" else (2500.0 / 3 if dname == "f32x3" else 2500.0)        # x3: three bf16 MFMAs per product
            print(json.dumps({"what": "scoring (embed_dataset)", "dtype": dname, "batch_size": bs, "masked_batches": bool(masked and bs > 1), "utterances": args.n,
                              "mean_seconds_per_utt": round(audio_s / args.n, 2), "utt_per_s": round(args.n / dtm, 2), "audio_s_per_s": round(audio_s / dtm, 1),
                              "frontend_tflops": round(flops / dtm / 1e12, 2), "peak_tflops": peak, "frac_of_peak": round(flops / dtm / 1e12 / peak, 4),
                              "max_abs_emb_diff_vs_batch1": float((emb - base).abs().max())}), flush=True)
    del model
    torch.cuda.empty_cache()
