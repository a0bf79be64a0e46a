"""GPU time of the graph-attention layers of the AASIST back-end at bs 64 (forward + backward, events on the stream): how much of the
back-end's ~4 ms they are.  usage: python scripts/time_gat.py [bs]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd.models.sslassist import AasistBackend

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
be = AasistBackend(device="cuda", seed=0, compute=os.environ.get("GAT_COMPUTE", "bf16"))
g = torch.Generator().manual_seed(0)
def timed(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = 0.0
for pre, N in (("GAT_layer_S", 42), ("GAT_layer_T", 66)):
    x = torch.randn(B, N, 64, generator=g).cuda(); dout = torch.randn(B, N, 64, generator=g).cuda()
    c = {}
    def f():
        be._site_id = 0
        be._gat_fwd(pre, x, B, N, c, True, {})
    def fb():
        f(); be._gat_bwd(pre, dout, B, N, c)
    tf, tfb = timed(f), timed(fb)
    tot += tfb
    print("%-22s N=%3d  fwd %7.1f us   fwd+bwd %7.1f us" % (pre, N, tf, tfb))
for pre, N1, N2, Din in (("HtrgGAT_layer_ST11", 33, 21, 64), ("HtrgGAT_layer_ST12", 16, 10, 32)):
    x1 = torch.randn(B, N1, Din, generator=g).cuda(); x2 = torch.randn(B, N2, Din, generator=g).cuda()
    master = be.p["master1"] if Din == 64 else torch.randn(B, 32, generator=g).cuda()
    d1, d2, dm = torch.randn(B, N1, 32, generator=g).cuda(), torch.randn(B, N2, 32, generator=g).cuda(), torch.randn(B, 32, generator=g).cuda()
    dmaster = torch.zeros(B, Din, device="cuda")
    c = {}
    def f():
        be._site_id = 0
        be._htrg_fwd(pre, x1, x2, master, 0 if Din == 64 else 32, B, N1, N2, Din, c, True, {})
    def fb():
        f(); be._htrg_bwd(pre, d1, d2, dm, dmaster, Din, B, c)
    tf, tfb = timed(f), timed(fb)
    tot += 2 * tfb
    print("%-22s N=%3d  fwd %7.1f us   fwd+bwd %7.1f us  (x2 per step)" % (pre, N1 + N2, tf, tfb))
print("graph-attention layers per step: %.2f ms" % (tot / 1e3))
