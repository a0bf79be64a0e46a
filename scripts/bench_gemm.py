"""Micro-benchmark of occ_gemm on the front-end shapes (bf16): kernel variants A/B'd in ONE process, interleaved rounds,
random operands.  usage: bench_gemm.py [variants, comma separated; default "1,5"] [rounds]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import ops
from occm_amd._lib import lib

variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,5").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
shapes = [("fc1", 6368, 4096, 1024, True), ("fc2", 6368, 1024, 4096, False), ("qkv", 6368, 3072, 1024, False), ("out", 6368, 1024, 1024, False),
          ("conv1", 204768, 512, 1536, False), ("conv3", 51168, 512, 1536, False), ("conv5", 12768, 512, 1024, False), ("sq4k", 4096, 4096, 4096, False)]
if os.environ.get("GEMM_SHAPES"):
    shapes = [("s%d" % i, *[int(v) for v in t.split("x")], False) for i, t in enumerate(os.environ["GEMM_SHAPES"].split(","))]
g = torch.Generator().manual_seed(0)
print("variants", variants)
for name, M, N, K, gelu in shapes:
    x = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    act = ops.ACT_GELU if gelu else ops.ACT_NONE
    ref = torch.nn.functional.linear(x.float(), w.float(), b)
    ref = torch.nn.functional.gelu(ref) if gelu else ref
    best, err = {v: [] for v in variants}, {}
    for v in variants:
        lib().occ_gemm_variant(v)
        out.zero_()
        for _ in range(2):
            ops.linear(x, w, b, act=act, out=out)
        err[v] = float((out.float() - ref).abs().max())
    n = 10
    for r in range(rounds):
        for v in variants:
            lib().occ_gemm_variant(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                ops.linear(x, w, b, act=act, out=out)
            e1.record(); torch.cuda.synchronize()
            best[v].append(e0.elapsed_time(e1) / n)
    line = "%-6s M=%6d N=%5d K=%5d " % (name, M, N, K)
    if os.environ.get("GEMM_VENDOR"):                  # yardstick only: torch's bf16 addmm (hipBLASLt) on the same operands, no GELU
        bb = b.bfloat16(); wt = w.t()
        vt = []
        for r in range(rounds + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                torch.addmm(bb, x, wt, out=out)
            e1.record(); torch.cuda.synchronize()
            vt.append(e0.elapsed_time(e1) / n)
        vt = sorted(vt[1:])
        line += " | vendor %7.1f us %6.0f TF" % (vt[len(vt) // 2] * 1e3, 2 * M * N * K / vt[len(vt) // 2] / 1e9)
    for v in variants:
        t = sorted(best[v]); med = t[len(t) // 2]
        line += " | v%d %7.1f us %6.0f TF (min %6.0f) err %.2g" % (v, med * 1e3, 2 * M * N * K / med / 1e9, 2 * M * N * K / t[0] / 1e9, err[v])
    print(line, flush=True)
lib().occ_gemm_variant(1)
