"""Training soak (scripts/soak_train.py) as a test: 160 optimizer steps of the configs[2] step (XLS-R-300M fine-tuned end to end + AASIST,
RawBoost 5, bf16) at bs 24 on a synthetic task with a learnable signal and fresh waveforms every step, in a child process.  The
descriptiveness loss of the last tenth must be below 0.8x the first tenth's (measured at bs 48 / 1000 steps: 0.76 -> 0.048, fp8 0.67 -> 0.055,
profiles/r03_soak_train*.json), every loss finite, and the allocator's high-water mark after 20 steps is the one at the end."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_finetuning_learns_a_synthetic_task_and_stays_finite():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "soak_train.py"), "160", "24"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["all_finite"] and out["loss_d_last_tenth"] < 0.8 * out["loss_d_first_tenth"], out
    assert out["max_mem_GB_end"] <= out["max_mem_GB_after_20_steps"] * 1.001, out
