import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): built on demand from oracle/*.c."""
    import subprocess
    so = os.path.join(REPO, "oracle", "libpolar_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")])
    from oracle import oracle_py
    return oracle_py


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    return {k: d[k] for k in d.files}


def unpack_bits(words, N):
    """[B][N/32] uint32 words -> [B][N] 0/1"""
    w = np.ascontiguousarray(words).view(np.uint32).reshape(-1, N // 32)
    return ((w[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1).reshape(-1, N).astype(np.int32)
