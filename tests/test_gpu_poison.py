"""Uninitialised-read screen (scripts/poison_check.py) as a test: in a child process every torch.empty buffer is NaN-filled before the
library sees it; three training steps (fine-tuning in bf16 at head sizes 64 and 80, fine-tuning with fp8 operands, frozen front-end) and a
scoring forward must stay finite.  A fresh process sees zero-filled pages from the driver, so a kernel that reads an element nobody wrote
passes every single-call test and fails only in a long-running job."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_no_kernel_reads_uninitialised_buffers():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "poison_check.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("-> ok") == 5, r.stdout
