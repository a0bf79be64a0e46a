#!/usr/bin/env python3
"""Headline benchmark: utterances/s of one training step on 4 s @ 16 kHz synthetic utterances.

Workload at N=1 = BASELINE.json configs[1]: XLS-R-300M frozen front-end (bf16 MFMA) + AASIST back-end
(fwd + loss + bwd + Adam, f32), bs=32 per GPU.  N>1: weak scaling, 32 utterances per rank, one process per GPU
(torchrun), RCCL all-reduce of the flat back-end gradient.  One JSON line on rank 0.

    python bench.py --gpus 1 --steps 20 --warmup 5
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

L_SAMPLES, BS = 64000, 32
BF16_DENSE_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA


def synth_batch(B, rank, device):
    g = torch.Generator().manual_seed(1234 + rank)
    wav = (0.1 * torch.randn(B, L_SAMPLES, generator=g)).clamp(-1, 1)
    labels = (torch.arange(B) % 12 >= 6).long()              # [0]*6 + [1]*6 groups (oc_training.py:215-240)
    return wav.to(device), labels.to(device)


def gemm_flops_per_utt(cfg, L):
    """Algorithmic FLOPs of the launches of the bf16 GEMM kernel per utterance (SURVEY.md section 8d formula terms
    FE (layers 1-6) + PROJ + POS + LIN); layer 0 of the conv stack and attention run in other kernels."""
    Ts, Lc = [], L
    for k, s in [(10, 5)] + [(3, 2)] * 4 + [(2, 2)] * 2:
        Lc = (Lc - k) // s + 1
        Ts.append(Lc)
    C, T, d, f, n = 512, Ts[-1], cfg.dim, cfg.ffn, cfg.layers
    fe = 2 * (sum(Ts[1:5]) * C * C * 3 + sum(Ts[5:7]) * C * C * 2)
    proj = 2 * T * C * d
    pos = 2 * (T + 1) * d * (d // cfg.pos_groups) * cfg.pos_k
    lin = 2 * T * (4 * d * d + 2 * d * f) * n
    return fe + proj + pos + lin


def usable_cores():
    """Cores this job may really use: the cgroup CPU quota when there is one (a 1-GPU box exposes all host CPUs but grants a
    share), else the affinity mask; never more than 64 threads (the oracle's small ops stop scaling long before)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except (OSError, ValueError, IndexError):
            pass
    n = min(n, int(os.environ.get("OCC_CPU_THREADS", "64")))
    return max(1, n)


def gemm_bytes_per_step(cfg, L, B):
    """Algorithmic HBM bytes of the bf16 GEMM launches of one step: each launch reads A and W once and writes C once."""
    Ts, Lc = [], L
    for k, s in [(10, 5)] + [(3, 2)] * 4 + [(2, 2)] * 2:
        Lc = (Lc - k) // s + 1
        Ts.append(Lc)
    T, d, f, n = Ts[-1], cfg.dim, cfg.ffn, cfg.layers
    M = B * T
    tot = 0
    for i in range(1, 7):
        k = 3 if i < 5 else 2
        tot += B * Ts[i - 1] * 512 * 2 + 512 * k * 512 * 2 + B * Ts[i] * 512 * 2
    tot += M * 512 * 2 + d * 512 * 2 + M * d * 2                                   # post_extract_proj
    tot += B * (T + cfg.pos_k) * d * 2 + d * (d // cfg.pos_groups) * cfg.pos_k * 2 + M * d * 4 + M * d * 2   # pos conv (+residual)
    per_layer = (M * d * 2 + 3 * d * d * 2 + M * 3 * d * 2) + (M * d * 2 + d * d * 2 + 2 * M * d * 4) \
        + (M * d * 2 + f * d * 2 + M * f * 2) + (M * f * 2 + d * f * 2 + 2 * M * d * 4)
    return tot + n * per_layer


def cpu_baseline(budget_s=25.0):
    """The torch-CPU oracle (kind "port": the reference's fairseq front-end cannot run) on a bounded sample of the same
    workload: frozen XLS-R-300M forward + AASIST fwd/bwd + Adam, bs=2, as many steps as fit the budget (>= 1)."""
    from oracle import aasist_ref, losses_ref, xlsr_ref
    from oracle.fill import fill_like
    cores = usable_cores()
    torch.set_num_threads(cores)
    cfg = xlsr_ref.XlsrConfig.xlsr_300m()
    px = fill_like(xlsr_ref.param_shapes(cfg), seed=0)
    pb = fill_like(aasist_ref.param_shapes(), seed=0)
    train = [v.requires_grad_(True) for k, v in pb.items() if v.dtype.is_floating_point and not k.split(".")[-1].startswith("running")]
    opt = torch.optim.Adam(train, lr=1e-5)
    B = 2
    wav, labels = synth_batch(B, 0, "cpu")
    t0 = time.time()
    steps = 0
    while steps < 1 or (time.time() - t0) < budget_s * 0.6:
        with torch.no_grad():
            feats = xlsr_ref.extract_feat(wav, px, cfg)
        opt.zero_grad()
        emb, out = aasist_ref.backend_forward(feats, pb, train=True)
        loss = 0.0 * losses_ref.compactness_loss(emb) + 1.0 * losses_ref.descriptiveness_loss(out, labels)
        loss.backward()
        opt.step()
        steps += 1
    dt = time.time() - t0
    return {"value": round(B * steps / dt, 4), "unit": "utterances/s", "cores": cores, "kind": "port",
            "sample": "%d step(s) of bs=%d, fp32 torch-CPU oracle (XLS-R-300M frozen fwd + AASIST fwd/bwd + Adam), %d threads" % (steps, B, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="do not replay the frozen front-end from a HIP graph")
    ap.add_argument("--finetune", nargs="?", const="full", default=None, choices=["full", "encoder"],
                    help="NOT the headline config: also train XLS-R (full = end-to-end as the reference's optimizer does, BASELINE configs[2] "
                    "minus RawBoost; encoder = transformer only); prints the same JSON with a different workload name")
    ap.add_argument("--bs", type=int, default=BS, help="per-GPU batch (headline: 32)")
    ap.add_argument("--rawboost", type=int, default=0, help="RawBoost algo 1-8 applied on the GPU inside the timed step (configs[2]: 5)")
    ap.add_argument("--xlsr", default="300m", choices=["300m", "1b"], help="NOT the headline config with 1b: XLS-R-1B geometry (48 layers, d 1280, heads of 80; "
                    "BASELINE configs[4] asks for it in fp8 over 8 GPUs -- this runs it in bf16)")
    ap.add_argument("--backend", default="aasist", choices=["aasist", "senet"], help="NOT the headline config with senet: SE-ResNet34 on the XLS-R "
                    "features (models/senet.py ssl_resnet34, loss 0.1 c + 0.9 d as test_dataloader_v2.py:127)")
    ap.add_argument("--no-overlap", action="store_true", help="do not compute the next batch's frozen-front-end features on a side stream "
                    "while the back-end trains on the current one")
    ap.add_argument("--split", type=int, default=1, help="run the front-end as this many concurrent sub-batches on separate HIP "
                    "streams inside the graph; measured slower on MI355X (16.4 / 19.8 / 21.9 ms per step at 1 / 2 / 4), kept for experiments")
    args = ap.parse_args()

    from occm_amd import ops, parallel
    from occm_amd._lib import require_gpu
    from occm_amd.models import xlsr
    from occm_amd.models.sslassist import AModel
    from occm_amd.trainer import OcTrainer
    require_gpu()
    rank, world, local = parallel.init_from_env()
    local = local % max(1, torch.cuda.device_count())       # (rehearsals with more ranks than GPUs share a device)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = xlsr.XlsrConfig.xlsr_300m() if args.xlsr == "300m" else xlsr.XlsrConfig.xlsr_1b()
    bs = args.bs
    if args.backend == "senet":
        from occm_amd.models.senet import ssl_resnet34
        model = ssl_resnet34(dev, ssl_cfg=cfg, ssl_dtype=torch.bfloat16, finetune_ssl=args.finetune or False)
        wc, wd = 0.1, 0.9
    else:
        model = AModel(None, dev, ssl_cfg=cfg, ssl_dtype=torch.bfloat16, seed=0, finetune_ssl=args.finetune or False)
        wc, wd = 0.0, 1.0
    model.train()
    trainer = OcTrainer(model, lr=1e-5, w_compact=wc, w_descr=wd, train_frontend=bool(args.finetune), rawboost_algo=args.rawboost, group_size=12 if bs % 12 == 0 else None)
    wav, labels = synth_batch(bs, rank, dev)
    fe = model.ssl_model.model
    if args.finetune:
        args.no_graph = True

    # The frozen front-end is a fixed launch sequence: capture it once into a HIP graph (launch-bound otherwise).
    graph = None
    if not args.no_graph:
        static_wav = wav.clone()
        orig_forward = fe.forward
        nsplit = max(1, args.split)
        T = xlsr.n_frames(L_SAMPLES)
        static_feats = torch.empty(bs, T, cfg.dim, device=dev, dtype=torch.float32)
        cuts = [bs * i // nsplit for i in range(nsplit + 1)]
        side = [torch.cuda.Stream(device=dev) for _ in range(nsplit - 1)]

        def split_forward():
            main = torch.cuda.current_stream()
            for i in range(nsplit):
                st = main if i == 0 else side[i - 1]
                if st is not main:
                    st.wait_stream(main)
                with torch.cuda.stream(st):
                    orig_forward(static_wav[cuts[i]:cuts[i + 1]], out_dtype=torch.float32, slot=i, out=static_feats[cuts[i]:cuts[i + 1]])
            for st in side:
                main.wait_stream(st)
        split_forward()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # thread_local: with world > 1 the RCCL watchdog thread polls events while this thread captures
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            split_forward()

        def replay_forward(w, out_dtype=None, taps=None):
            static_wav.copy_(w)
            graph.replay()
            return static_feats
        fe.forward = replay_forward

    # Every step does the whole path on one batch.  With the frozen front-end the trainer software-pipelines two steps: the data
    # stage (RawBoost + XLS-R features) of step i+1 runs on a side stream while the back-end of step i trains; the timed region
    # still contains exactly `steps` data stages and `steps` back-end updates (the first data stage is not overlapped).
    overlap = not args.no_overlap and not args.finetune
    for i in range(args.warmup):
        trainer.step(wav, labels, next_wav=wav if overlap and i + 1 < args.warmup else None)
    parallel.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        trainer.step(wav, labels, next_wav=wav if overlap and i + 1 < args.steps else None)
    torch.cuda.synchronize(); parallel.barrier()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    lc, ld = trainer.last
    loss_d = float(ld.item())

    # ---- roofline of the dominant kernel (bf16 MFMA GEMM): algorithmic FLOPs / measured launch time ----
    roof = None
    if rank == 0 and not args.finetune:
        if graph is not None:
            fe.forward = orig_forward
        ops.PROFILE = []
        for _ in range(3):
            fe.forward(wav, out_dtype=torch.float32)
        torch.cuda.synchronize()
        recs, ops.PROFILE = ops.PROFILE, None
        t_ms = sum(a.elapsed_time(b) for kind, a, b in recs if kind == "gemm_bf16") / 3.0
        n_launch = sum(1 for kind, _, _ in recs if kind == "gemm_bf16") // 3
        fl = gemm_flops_per_utt(cfg, L_SAMPLES) * bs
        ach = fl / (t_ms * 1e-3) / 1e12
        traffic, tnote = None, None
        try:      # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950-corrected)
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm.json")))
            ks = [v for k, v in pm["kernels"].items() if "gemm_bf16_" in k]          # the bf16 GEMM family (default + half-slab kernels)
            traffic = int(sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for v in ks) / sum(v["launches"] for v in ks))
            tnote = "profiles/r01_pmc_hbm.json (separate --pmc passes of this command; 2*FETCH_SIZE + WRITE_SIZE)"
        except (OSError, KeyError, ValueError):
            pass
        roof = {"kernel": "bf16 GEMM (gemm_bf16_dma_kernel, its in-workgroup split-K form gemm_bf16_ks2_kernel and the 128x64 form for the grouped conv: all front-end Linear/Conv1d launches of one step)", "bound": "mfma",
                "achieved": round(ach, 1), "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / BF16_DENSE_PEAK_TFLOPS, 4),
                "traffic": traffic, "traffic_source": tnote, "algorithmic_bytes_per_launch": int(gemm_bytes_per_step(cfg, L_SAMPLES, bs) / max(n_launch, 1)),
                "launches_per_step": n_launch, "avg_launch_us": round(t_ms * 1e3 / max(n_launch, 1), 2), "flops_per_step": fl}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
    if rank == 0:
        if args.backend == "senet":
            wl = "XLSR-%s %s frontend + SE-ResNet34 backend, bs=%d per GPU (not a BASELINE config; models/senet.py ssl_resnet34)" % (args.xlsr.upper(), "fine-tuned" if args.finetune else "frozen", bs)
        else:
            wl = ("XLSR-300M frozen frontend + AASIST backend, bs=%d per GPU, 64000-sample utterances (BASELINE configs[1])" % bs) if not args.finetune else \
            ("XLSR-300M fine-tuned (%s) + AASIST backend, bs=%d per GPU (BASELINE configs[2]%s)" % (args.finetune, bs, ", RawBoost algo %d on-GPU" % args.rawboost if args.rawboost else " without RawBoost"))
        out = {"metric": "utterances/sec (4 s @16 kHz) training step", "value": round(bs * world * args.steps / dt, 2), "unit": "utterances/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": wl,
                          "global_batch": bs * world, "samples_per_utt": L_SAMPLES, "parallelism": "dp%d" % world,
                          "frontend": "bf16 MFMA, f32 accumulate, HIP-graph replay" + (", features of step i+1 computed on a side stream during step i's back-end" if overlap else ""), "backend": ("fwd+bwd, f32 storage, bf16-MFMA GEMMs and weight gradients (f32 accumulate), dropout on, Adam lr=1e-5" if args.backend == "aasist"
                                      else "SE-ResNet34 fwd+bwd, f32 storage, bf16-MFMA convolutions and weight gradients (f32 accumulate), Adam lr=1e-5"),
                          "loss": "%.1f*compactness + %.1f*descriptiveness (%s)" % (wc, wd, "oc_training.py:380-381" if args.backend == "aasist" else "test_dataloader_v2.py:127"), "final_loss_d": round(loss_d, 5)},
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
