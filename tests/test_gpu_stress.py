"""GPU: the differential sweep of tools/stress_parity.py as a test -- 240 shapes (SCL / CA-SCL with N = 32..4096, L = 1..32,
three rates, CRC-6 / CRC-24C / none; SC below and above the 64-frame batch threshold; BP), f64 and f32, every one
compared with the CPU oracle through the C ABI.  A few seconds on the GPU; the oracle side dominates."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_differential_sweep_all_shapes_identical():
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "stress_parity.py")], capture_output=True, text=True,
                         timeout=1500)
    tail = out.stdout[-3000:] + out.stderr[-2000:]
    assert out.returncode == 0, tail
    lines = [l for l in out.stdout.splitlines() if l.startswith(("ok ", "BAD"))]
    assert len(lines) >= 238 and not any(l.startswith("BAD") for l in lines), tail
    assert "0 mismatching configurations" in out.stdout
