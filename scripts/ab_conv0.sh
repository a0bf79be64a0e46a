#!/bin/bash
# conv0 forward: VALU kernel vs the matrix-core kernel (one process per setting), B = 32 and a max-abs difference check
for m in 0 1; do
  echo "== OCC_CONV0_MFMA=$m"; OCC_CONV0_MFMA=$m python scripts/bench_conv0.py 2>&1 | grep torch
done
python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
import subprocess
code = '''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from occm_amd import ops
g = torch.Generator().manual_seed(3)
wav = (0.1 * torch.randn(3, 20003, generator=g)).cuda()
w = (torch.randn(512, 10, generator=g) * 0.3).cuda(); b = torch.randn(512, generator=g).cuda() * 0.1
ga = (1 + 0.1 * torch.randn(512, generator=g)).cuda(); be = (0.1 * torch.randn(512, generator=g)).cuda()
out = ops.conv0_ln_gelu(wav, w, b, ga, be, 10, 5, out_dtype=torch.bfloat16)
ref = ops.conv0_ln_gelu(wav, w, b, ga, be, 10, 5, out_dtype=torch.float32)
d = (out.float() - ref).abs()
print("bf16 out vs f32 kernel: max abs diff %.4e (|ref| max %.3f), frac of elements off by more than one bf16 ulp: %.5f" % (float(d.max()), float(ref.abs().max()), float((d > 0.0079 * ref.abs().clamp_min(0.01)).float().mean())))
'''
for m in ("0", "1"):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OCC_CONV0_MFMA=m), capture_output=True, text=True)
    print("OCC_CONV0_MFMA=" + m, r.stdout.strip(), r.stderr.strip()[-300:])
PY
