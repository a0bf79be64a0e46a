"""Would the weight-gradient launches of the backward pass fill the idle CUs of the input-gradient chain if they ran on a second stream?
One transformer layer's backward GEMMs at bs 64 (M = 12736) -- the four input gradients (912 / 228 / 228 / 228 workgroups: 11 % of the
CUs idle in their last or only round) and the two paired weight gradients (256 workgroups) -- times 12, once on one stream in program
order, once with the pairs on a side stream (no dependencies between the two: operands are separate buffers)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import ops, backend_ops as K
from occm_amd._lib import ACT_MUL_AUX, OCC_BF16
from occm_amd.ops import rowmap

M, D, F = 12736, 1024, 4096
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).bfloat16().cuda()
dy, du, dh, da, dqkv = r(M, D), torch.empty(M, F, device="cuda", dtype=torch.bfloat16), torch.empty(M, D, device="cuda", dtype=torch.bfloat16), \
    torch.empty(M, D, device="cuda", dtype=torch.bfloat16), r(M, 3 * D)
aux, f, h2, att, h1 = r(M, F), r(M, F), r(M, D), r(M, D), r(M, D)
w2T, w1T, woT, wqT = r(F, D, sc=D ** -0.5), r(D, F, sc=F ** -0.5), r(D, D, sc=D ** -0.5), r(D, 3 * D, sc=D ** -0.5)
du_w = r(M, F)                       # the pairs read their own copies: no hazards between the streams in this experiment
gw2, gw1, gwo, gwq = [torch.zeros(*s, device="cuda") for s in ((D, F), (F, D), (D, D), (3 * D, D))]
gb = [torch.zeros(n, device="cuda") for n in (D, F, D, 3 * D)]


def dgrads():
    ops.gemm_raw(M, F, D, dy, rowmap(M, 0, D), w2T, D, du, rowmap(M, 0, F), OCC_BF16, OCC_BF16, act=ACT_MUL_AUX, aux=aux)
    ops.gemm_raw(M, D, F, du, rowmap(M, 0, F), w1T, F, dh, rowmap(M, 0, D), OCC_BF16, OCC_BF16)
    ops.gemm_raw(M, D, D, dy, rowmap(M, 0, D), woT, D, da, rowmap(M, 0, D), OCC_BF16, OCC_BF16)
    ops.gemm_raw(M, D, 3 * D, dqkv, rowmap(M, 0, 3 * D), wqT, 3 * D, dh, rowmap(M, 0, D), OCC_BF16, OCC_BF16)


def pairs():
    K.gemm_tn_pair(M, (D, F, dy, rowmap(M, 0, D), f, rowmap(M, 0, F), gw2, F, gb[0]), (F, D, du_w, rowmap(M, 0, F), h2, rowmap(M, 0, D), gw1, D, gb[1]))
    K.gemm_tn_pair(M, (D, D, dy, rowmap(M, 0, D), att, rowmap(M, 0, D), gwo, D, gb[2]), (3 * D, D, dqkv, rowmap(M, 0, 3 * D), h1, rowmap(M, 0, D), gwq, D, gb[3]))


side = torch.cuda.Stream()
REP = 12


def serial():
    for _ in range(REP):
        dgrads(); pairs()


def two():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    for _ in range(REP):
        dgrads()
        with torch.cuda.stream(side):
            pairs()
    main.wait_stream(side)


for name, fn in (("one stream", serial), ("two streams", two), ("one stream", serial), ("two streams", two)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / REP * 1e3)
    ts.sort()
    print("%-12s %8.1f us per layer (median of 5; min %.1f)" % (name, ts[2], ts[0]), flush=True)
