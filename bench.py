#!/usr/bin/env python3
"""Headline benchmark: utterances/s of one training step on 4 s @ 16 kHz synthetic utterances.

Workload at N=1 = BASELINE.json configs[2], the largest single-GPU configuration: XLS-R-300M fine-tuned end to end (every
parameter in the optimizer, as oc_training.py:324) + AASIST back-end, bs=64, RawBoost algo 5 on the GPU inside the step.
N>1 = configs[3]: the same step on every rank (64 utterances per GPU, weak scaling), one process per GPU, per-layer RCCL
all-reduce of the 1.26 GB flat gradient overlapped with backward.  `--frozen` runs configs[1] (frozen front-end, bs=32).
One JSON line on rank 0.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus 8            # starts 8 ranks itself (torch.distributed.run) when WORLD_SIZE is not set
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

L_SAMPLES = 64000
BS_FINETUNE, BS_FROZEN = 64, 32
BF16_DENSE_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
GEMM_SOURCES = ("gemm.hip", "gemm_common.h", "gemm_tn.hip", "gemm_p8.hip", "gemm_q4.hip", "gemm_tn_p8.hip")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frozen", action="store_true", help="BASELINE configs[1] instead of the headline: XLS-R frozen (HIP-graph replay), bs 32, no RawBoost")
    ap.add_argument("--finetune", nargs="?", const="full", default=None, choices=["full", "encoder"],
                    help="full (default workload) = end-to-end as the reference's optimizer does; encoder = transformer only (not a BASELINE config)")
    ap.add_argument("--bs", type=int, default=None, help="per-GPU batch (headline: 64; --frozen: 32)")
    ap.add_argument("--rawboost", type=int, default=None, help="RawBoost algo 0-8 on the GPU inside the timed step (headline: 5; --frozen: 0)")
    ap.add_argument("--xlsr", default="300m", choices=["300m", "1b"], help="1b: XLS-R-1B geometry (48 layers, d 1280, heads of 80), not the headline")
    ap.add_argument("--backend", default="aasist", choices=["aasist", "senet"], help="senet: SE-ResNet34 on the XLS-R features (not the headline)")
    ap.add_argument("--no-graph", action="store_true", help="--frozen only: do not replay the frozen front-end from a HIP graph")
    ap.add_argument("--no-overlap", action="store_true", help="no side-stream prefetch of the next batch's data stage (frozen: RawBoost + features; fine-tuned: RawBoost)")
    ap.add_argument("--fp8", action="store_true", help="not the headline: forward / input-gradient GEMMs of the transformer layers on the fp8 MFMA path (e4m3 / e5m2, "
                    "delayed per-tensor scaling), as BASELINE configs[4] asks for XLS-R-1B; weight gradients stay bf16")
    ap.add_argument("--grad-wire", default="f32", choices=["f32", "bf16"], help="N > 1: dtype of the XLS-R gradients on the xGMI links (bf16 halves the 1.26 GB "
                    "payload; sums are widened back into the f32 gradient buffer before Adam).  Default f32 = the exact sum")
    ap.add_argument("--ssl-dropouts", default=None, metavar="P_RES,P_ATT,P_ACT,P_LAYERDROP",
                    help="not the headline: fairseq's train-mode probabilities of the fine-tuned front-end (dropout, attention_dropout, activation_dropout, "
                         "encoder_layerdrop), e.g. 0.1,0.1,0.1,0.05 -- the published XLS-R pre-training configuration the headline uses has all of them 0; a "
                         "fine-tuning checkpoint's cfg may not (the reference keeps XLS-R in train mode, oc_training.py:352)")
    ap.add_argument("--rccl-channels", type=int, default=0, help="N > 1: cap RCCL at this many channels (NCCL_MAX_NCHANNELS); 0 = RCCL's own choice (default)")
    ap.add_argument("--print-workload", action="store_true", help="print the workload tag of these flags (what scripts/pmc_summary.py stores in a PMC profile) and exit")
    ap.add_argument("--dry-launch", action="store_true", help="rendezvous only (gloo, no GPU call): every rank reports world size and its shard of the "
                    "utterance groups, rank 0 prints them as one JSON line")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------- launcher
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, args_ns=None):
    """--gpus N without a torchrun environment: start N ranks as children (one per GPU) BEFORE this process touches a GPU, pass their
    output through and exit with their status.  Replaces nn.DataParallel's single process (oc_training.py:328)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # RCCL's channel count is left alone: capping it (NCCL_MAX_NCHANNELS) would keep the collective inside the CUs the one-round GEMM
    # launches leave free, but it can also cap xGMI all-reduce bandwidth for the 1.26 GB exchange, and neither effect has been measured on
    # an 8-GPU node.  --rccl-channels N (or the caller's own NCCL_MAX_NCHANNELS) opts in; the value in force is printed in the bench line.
    if getattr(args_ns, "rccl_channels", 0):
        env["NCCL_MAX_NCHANNELS"] = str(args_ns.rccl_channels)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def dry_launch(args):
    """No GPU call: init_process_group over gloo, agree on the world size, shard the global batch's groups of 12."""
    import torch.distributed as dist
    from occm_amd import parallel
    rank, world, local = parallel.init_from_env(backend="gloo")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the process group has %d ranks" % (args.gpus, world))
    n_groups = 2 * world                                   # 2 groups of 12 per rank in the rehearsal
    lo, hi = parallel.shard_groups(n_groups, rank, world)
    mine = {"rank": rank, "local_rank": local, "world": world, "groups": [lo, hi]}
    if world > 1:
        seen = [None] * world
        dist.all_gather_object(seen, mine)
    else:
        seen = [mine]
    ok = all(s["world"] == world for s in seen) and sorted(s["rank"] for s in seen) == list(range(world))
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "ranks": seen, "n_groups": n_groups, "ok": ok}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


# ------------------------------------------------------------------------------------------------- workload arithmetic
def synth_batch(B, rank, device):
    import torch
    g = torch.Generator().manual_seed(1234 + rank)
    wav = (0.1 * torch.randn(B, L_SAMPLES, generator=g)).clamp(-1, 1)
    labels = (torch.arange(B) % 12 >= 6).long()              # [0]*6 + [1]*6 groups (oc_training.py:215-240)
    return wav.to(device), labels.to(device)


def conv_frames(L):
    Ts, Lc = [], L
    for k, s in [(10, 5)] + [(3, 2)] * 4 + [(2, 2)] * 2:
        Lc = (Lc - k) // s + 1
        Ts.append(Lc)
    return Ts


def gemm_flops_per_utt(cfg, L):
    """Algorithmic forward FLOPs of the bf16 GEMM launches per utterance (SURVEY.md section 8d formula terms FE (layers 1-6) + PROJ +
    POS + LIN); layer 0 of the conv stack and attention run in other kernels.  Fine-tuning runs each of them three times
    (forward, input gradient, weight gradient)."""
    Ts = conv_frames(L)
    C, T, d, f, n = 512, Ts[-1], cfg.dim, cfg.ffn, cfg.layers
    fe = 2 * (sum(Ts[1:5]) * C * C * 3 + sum(Ts[5:7]) * C * C * 2)
    proj = 2 * T * C * d
    pos = 2 * (T + 1) * d * (d // cfg.pos_groups) * cfg.pos_k
    lin = 2 * T * (4 * d * d + 2 * d * f) * n
    return fe + proj + pos + lin


def gemm_bytes_per_step(cfg, L, B, finetune):
    """Algorithmic HBM bytes of the bf16 GEMM launches of one step: each launch reads its two operands once and writes its result
    once (bf16 operands, bf16 or f32 results as the path stores them); fine-tuning adds the input-gradient and weight-gradient
    launches (dY and W in, dX out; dY and X in, f32 dW accumulated = read + written)."""
    Ts = conv_frames(L)
    T, d, f, n = Ts[-1], cfg.dim, cfg.ffn, cfg.layers
    M = B * T
    shapes = []                                            # (rows, N, K, bytes of C per element)
    for i in range(1, 7):
        k = 3 if i < 5 else 2
        shapes.append((B * Ts[i], 512, k * 512, 2))
    shapes.append((M, d, 512, 2))
    tot = 0
    for (m, nn, kk, ce) in shapes:
        tot += m * kk * 2 + nn * kk * 2 + m * nn * ce
        if finetune:
            tot += (m * nn * 2 + nn * kk * 2 + m * kk * 2) + (m * nn * 2 + m * kk * 2 + 2 * nn * kk * 4)
    G, cg = cfg.pos_groups, d // cfg.pos_groups
    pos_fwd = B * (T + cfg.pos_k) * d * 2 + d * cg * cfg.pos_k * 2 + M * d * 4 + M * d * 2
    tot += pos_fwd
    if finetune:
        tot += pos_fwd + (M * d * 2 + B * (T + cfg.pos_k) * d * 2 + 2 * d * cg * cfg.pos_k * 4)
    layer = [(M, 3 * d, d, 2, 0), (M, d, d, 4, 4), (M, f, d, 2, 0), (M, d, f, 4, 4)]     # (.., C bytes, residual bytes)
    per = 0
    for (m, nn, kk, ce, re) in layer:
        per += m * kk * 2 + nn * kk * 2 + m * nn * (ce + re)
        if finetune:
            per += (m * nn * 2 + nn * kk * 2 + m * kk * 2) + (m * nn * 2 + m * kk * 2 + 2 * nn * kk * 4)
    if finetune:
        per += 2 * M * f * 2          # fc1's epilogue also stores the bf16 pre-activation; fc2's input-gradient epilogue reads it (GELU')
    return tot + n * per


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


def cpu_baseline(finetune, rawboost, budget_s=25.0):
    """The torch-CPU oracle (kind "port": the reference's fairseq front-end cannot run) on a bounded sample of the same step at bs=8
    (SURVEY.md section 8d):
    (numpy RawBoost ->) XLS-R-300M (forward only when frozen, forward + backward when fine-tuned) -> AASIST fwd/bwd -> Adam over
    every trained parameter; as many steps as fit the budget (>= 1)."""
    import numpy as np
    import torch
    from oracle import aasist_ref, losses_ref, rawboost_np, xlsr_ref
    from oracle.fill import fill_like
    cores = usable_cores()
    torch.set_num_threads(cores)
    cfg = xlsr_ref.XlsrConfig.xlsr_300m()
    px = fill_like(xlsr_ref.param_shapes(cfg), seed=0)
    pb = fill_like(aasist_ref.param_shapes(), seed=0)
    train = [v.requires_grad_(True) for k, v in pb.items() if v.dtype.is_floating_point and not k.split(".")[-1].startswith("running")]
    if finetune:
        train += [v.requires_grad_(True) for v in px.values()]
    opt = torch.optim.Adam(train, lr=1e-5)
    B = 8
    wav, labels = synth_batch(B, 0, "cpu")
    rb_args = None
    if rawboost:
        from occm_amd.oc_training import rawboost_args
        rb_args = rawboost_args()
        np.random.seed(0)
    t0 = time.time()
    steps = 0
    while steps < 1 or (time.time() - t0) < budget_s * 0.6:
        x = wav
        if rawboost:
            x = torch.from_numpy(np.stack([rawboost_np.process_rawboost(w.numpy(), 16000, rb_args, rawboost) for w in wav]).astype(np.float32))
        if finetune:
            feats = xlsr_ref.extract_feat(x, px, cfg)
        else:
            with torch.no_grad():
                feats = xlsr_ref.extract_feat(x, px, cfg)
        opt.zero_grad()
        emb, out = aasist_ref.backend_forward(feats, pb, train=True)
        loss = 0.0 * losses_ref.compactness_loss(emb) + 1.0 * losses_ref.descriptiveness_loss(out, labels)
        loss.backward()
        opt.step()
        steps += 1
    dt = time.time() - t0
    what = "XLS-R-300M %s + AASIST fwd/bwd + Adam%s" % ("fwd/bwd (fine-tuned)" if finetune else "frozen fwd", ", numpy RawBoost algo %d" % rawboost if rawboost else "")
    return {"value": round(B * steps / dt, 4), "unit": "utterances/s", "cores": cores, "kind": "port",
            "sample": "%d step(s) of bs=%d, fp32 torch-CPU oracle (%s), %d threads" % (steps, B, what, cores)}


def gemm_source_sha():
    h = hashlib.sha256()
    for name in GEMM_SOURCES:
        path = os.path.join(ROOT, "occm_amd", "csrc", name)
        if os.path.exists(path):
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def workload_tag(args, finetune, bs, rawboost):
    """Identifies the measured workload (model, back-end, batch, dtype, augmentation): a PMC profile is quoted only for its own workload."""
    return "xlsr-%s/%s/%s/bs%d/rawboost%d/%s" % (args.xlsr, args.backend, ("finetune-" + str(finetune)) if finetune else "frozen", bs, rawboost, "fp8" if args.fp8 else "bf16") + \
        (("/drop" + args.ssl_dropouts) if getattr(args, "ssl_dropouts", None) else "")


def pmc_traffic(tag):
    """HBM bytes per GEMM launch from the committed PMC passes -- reported only when a profile of THIS workload (its tag is stored in the
    profile) taken with the GEMM sources of this tree (their hash is stored too) exists under profiles/; otherwise null: a counter from
    another model, dtype, batch or build must not look like a measurement of this run."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_*.json")), reverse=True):
        try:
            pm = json.load(open(path))
            if pm.get("workload") != tag or pm.get("gemm_src_sha16") != gemm_source_sha():
                continue
            ks = [v for k, v in pm["kernels"].items() if ("gemm_" in k and "bf16" in k) or "gemm_p8_kernel" in k or "gemm_tn_p8" in k or "gemm_tn_dma" in k]
            # the slab reduce that follows a weight-gradient kernel belongs to that launch: its bytes count, its launches do not
            red = [v for k, v in pm["kernels"].items() if "tn_p8_reduce" in k]
            n = sum(v["launches"] for v in ks)
            return int(sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for v in ks + red) / max(n, 1)), \
                "profiles/%s (separate --pmc passes of this command; 2*FETCH_SIZE + WRITE_SIZE, launch-weighted over the GEMM kernels, slab reduces included)" % os.path.basename(path)
        except (OSError, KeyError, ValueError):
            continue
    return None, "no PMC profile of this workload (%s) with these GEMM sources under profiles/" % tag


# ------------------------------------------------------------------------------------------------- main
def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], args))
    if args.dry_launch:
        sys.exit(dry_launch(args))
    if args.print_workload:
        ft = False if args.frozen else (args.finetune or "full")
        print(workload_tag(args, ft, args.bs if args.bs is not None else (BS_FROZEN if args.frozen else BS_FINETUNE),
                           args.rawboost if args.rawboost is not None else (0 if args.frozen else 5)))
        sys.exit(0)

    import torch
    from occm_amd import backend_ops, ops, parallel
    from occm_amd._lib import require_gpu
    from occm_amd.models import xlsr
    from occm_amd.models.sslassist import AModel
    from occm_amd.trainer import OcTrainer
    require_gpu()
    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the process group has %d ranks" % (args.gpus, world))
    local = local % max(1, torch.cuda.device_count())       # (rehearsals with more ranks than GPUs share a device)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = xlsr.XlsrConfig.xlsr_300m() if args.xlsr == "300m" else xlsr.XlsrConfig.xlsr_1b()
    finetune = False if args.frozen else (args.finetune or "full")
    bs = args.bs if args.bs is not None else (BS_FROZEN if args.frozen else BS_FINETUNE)
    rawboost = args.rawboost if args.rawboost is not None else (0 if args.frozen else 5)
    if args.backend == "senet":
        from occm_amd.models.senet import ssl_resnet34
        model = ssl_resnet34(dev, ssl_cfg=cfg, ssl_dtype=torch.bfloat16, finetune_ssl=finetune, synthetic_ssl=True)
        wc, wd = 0.1, 0.9
    else:
        model = AModel(None, dev, ssl_cfg=cfg, ssl_dtype=torch.bfloat16, seed=0, finetune_ssl=finetune, synthetic_ssl=True)
        wc, wd = 0.0, 1.0
    if args.fp8:
        if not finetune:
            raise SystemExit("bench.py: --fp8 applies to the fine-tuned front-end")
        model.ssl_model.model.enable_fp8()
    if args.ssl_dropouts:
        if not finetune:
            raise SystemExit("bench.py: --ssl-dropouts applies to the fine-tuned front-end (the frozen one runs its deterministic forward)")
        pr, pa, pc, pl = [float(v) for v in args.ssl_dropouts.split(",")]
        model.ssl_model.model.train_cfg = xlsr.XlsrTrainCfg(dropout=pr, attention_dropout=pa, activation_dropout=pc, encoder_layerdrop=pl)
    model.train()
    trainer = OcTrainer(model, lr=1e-5, w_compact=wc, w_descr=wd, train_frontend=bool(finetune), rawboost_algo=rawboost,
                        group_size=12 if bs % 12 == 0 else None, rank=rank, grad_wire_dtype=torch.bfloat16 if args.grad_wire == "bf16" else None)
    wav, labels = synth_batch(bs, rank, dev)
    fe = model.ssl_model.model

    # configs[1]: the frozen front-end is a fixed launch sequence: capture it once into a HIP graph (launch-bound otherwise).
    graph = None
    if not finetune and not args.no_graph:
        static_wav = wav.clone()
        orig_forward = fe.forward
        static_feats = torch.empty(bs, xlsr.n_frames(L_SAMPLES), cfg.dim, device=dev, dtype=torch.float32)
        orig_forward(static_wav, out_dtype=torch.float32, out=static_feats)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # thread_local: with world > 1 the RCCL watchdog thread polls events while this thread captures
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            orig_forward(static_wav, out_dtype=torch.float32, out=static_feats)

        def replay_forward(w, out_dtype=None, taps=None):
            static_wav.copy_(w)
            graph.replay()
            return static_feats
        fe.forward = replay_forward

    # Every step does the whole path on one batch.  With the frozen front-end the trainer software-pipelines two steps: the data
    # stage (RawBoost + XLS-R features) of step i+1 runs on a side stream while the back-end of step i trains; when the front-end is
    # trained only RawBoost of step i+1 can run ahead (under step i's back-end section).  Either way the timed region still contains
    # exactly `steps` data stages and `steps` updates (the first data stage is not overlapped).
    overlap = not args.no_overlap
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

    # ---- roofline of the dominant kernel family (bf16 MFMA GEMM): algorithmic FLOPs / launch time measured with HIP events on the
    # launch stream, in extra untimed steps of the same workload ----
    roof = None
    nprof = 2
    if finetune and rank != 0:
        # the extra steps contain the gradient all-reduce: every rank takes them, only rank 0 times its launches
        for _ in range(nprof):
            trainer.step(wav, labels)
        torch.cuda.synchronize()
    if rank == 0:
        if graph is not None:
            fe.forward = orig_forward
        ops.PROFILE = []
        if finetune:
            for _ in range(nprof):
                trainer.step(wav, labels)
        else:
            for _ in range(nprof):
                fe.forward(wav, out_dtype=torch.float32)
        torch.cuda.synchronize()
        recs, ops.PROFILE = ops.PROFILE, None
        sel = [(a, b) for kind, a, b in recs if kind == "gemm_bf16"]
        t_ms = sum(a.elapsed_time(b) for a, b in sel) / nprof
        n_launch = len(sel) // nprof
        fl = gemm_flops_per_utt(cfg, L_SAMPLES) * bs * (3 if finetune == "full" else 1)
        if finetune == "encoder":
            Ts = conv_frames(L_SAMPLES)
            fl = gemm_flops_per_utt(cfg, L_SAMPLES) * bs + 2 * (2 * Ts[-1] * (4 * cfg.dim ** 2 + 2 * cfg.dim * cfg.ffn) * cfg.layers) * bs
        ach = fl / (t_ms * 1e-3) / 1e12
        traffic, tnote = pmc_traffic(workload_tag(args, finetune, bs, rawboost))
        peak = 5000.0 if args.fp8 else BF16_DENSE_PEAK_TFLOPS       # (~5 PF dense fp8; the bf16 weight-gradient launches are priced against it too)
        roof = {"kernel": ("fp8 + " if args.fp8 else "") + "bf16 MFMA GEMM family: every Linear / Conv1d launch of the XLS-R front-end" +
                (" -- forward, input gradient (occ_gemm) and weight gradient (occ_gemm_tn)" if finetune else " (occ_gemm)"),
                "bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "traffic": traffic, "traffic_source": tnote,
                "algorithmic_bytes_per_launch": int(gemm_bytes_per_step(cfg, L_SAMPLES, bs, finetune == "full") / max(n_launch, 1)),
                "launches_per_step": n_launch, "avg_launch_us": round(t_ms * 1e3 / max(n_launch, 1), 2), "flops_per_step": fl,
                "gemm_ms_per_step": round(t_ms, 3)}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(bool(finetune), rawboost)
    if rank == 0:
        cfgno = "configs[1]" if not finetune else ("configs[2]" if world == 1 else "configs[3]")
        std = not args.fp8 and not args.ssl_dropouts and args.backend == "aasist" and args.xlsr == "300m" and finetune in (False, "full") and bs == (BS_FINETUNE if finetune else BS_FROZEN) and \
            rawboost == (5 if finetune else 0)
        wl = "XLSR-%s %s + %s backend, bs=%d per GPU, 64000-sample utterances%s (%s)" % (
            args.xlsr.upper(), ("fine-tuned end to end" if finetune == "full" else "fine-tuned (encoder only)") if finetune else "frozen frontend",
            "AASIST" if args.backend == "aasist" else "SE-ResNet34", bs, ", RawBoost algo %d on-GPU" % rawboost if rawboost else "",
            "BASELINE %s" % cfgno if std else "not a BASELINE config")
        out = {"metric": "utterances/sec (4 s @16 kHz) training step", "value": round(bs * world * args.steps / dt, 2), "unit": "utterances/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp8 (e4m3 / e5m2 transformer GEMMs) + bf16" if args.fp8 else "bf16", "data": "synthetic",
               "config": {"workload": wl, "global_batch": bs * world, "samples_per_utt": L_SAMPLES, "parallelism": "dp%d" % world,
                          "frontend": ("bf16 MFMA fwd + bwd, f32 accumulate, f32 master weights and gradients, Adam over all 315.4 M parameters" if finetune == "full" else
                                       "bf16 MFMA, f32 accumulate" + (", HIP-graph replay" if graph is not None else "") +
                                       (", features of step i+1 computed on a side stream during step i's back-end" if overlap else "")),
                          "frontend_train_mode": (repr(fe.train_cfg) + " (fairseq train-mode probabilities in force; the published XLS-R configuration has them all 0)"
                                                  if finetune else "frozen: deterministic forward, no dropout"),
                          "backend": ("fwd+bwd, f32 storage, bf16-MFMA GEMMs and weight gradients (f32 accumulate), dropout on, Adam lr=1e-5" if args.backend == "aasist"
                                      else "SE-ResNet34 fwd+bwd, f32 storage, bf16-MFMA convolutions and weight gradients (f32 accumulate), Adam lr=1e-5"),
                          "gradient_exchange": "none (1 rank)" if world == 1 else "RCCL all-reduce of the flat gradients (%s on the wire, f32 accumulation), per transformer layer, overlapped with backward; NCCL_MAX_NCHANNELS=%s" % (args.grad_wire, os.environ.get("NCCL_MAX_NCHANNELS", "default")),
                          "loss": "%.1f*compactness + %.1f*descriptiveness (%s)" % (wc, wd, "oc_training.py:380-381" if args.backend == "aasist" else "test_dataloader_v2.py:127"),
                          **({"configs4_note": "one GPU's shard (bs 32 of 256) of BASELINE configs[4]; measured, fp8 is 1-2 % ahead of bf16 at these widths (most of the "
                                               "fp8 GEMMs' saving is spent on quantisation passes and the weight gradients stay bf16: DESIGN.md section 5), within the "
                                               "pool's box-to-box spread: configs[4] is RUN IN BF16 by default and --fp8 is an accuracy-tested option"} if args.xlsr == "1b" else {}),
                          "final_loss_d": round(loss_d, 5), "gemm_src_sha16": gemm_source_sha()},
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
