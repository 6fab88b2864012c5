"""GPU: median ties (SCL_1024.c:619-633) and the kernels' two-step ranking, through the C ABI.

Fixtures tests/golden/ties_*.npz come from the COMPILED reference (see tests/test_oracle_ties.py for what they pin).
Every kernel that can decode a configuration is run on them:
  * POLAR_FLAG_TIE is raised on exactly the frames where the reference prints "Oops!" (or never returns);
  * decisions and path metric equal the REFERENCE's on every frame it returns from, tied or not, and the oracle's
    (the build's tie rule) on all frames;
  * POLAR_FLAG_RERANK -- the full-width re-rank behind the 32-bit pre-ranking -- is raised on exactly the frames
    where the oracle says the high words do not decide, including frames WITHOUT a tie, and the result is the
    reference's there too;
  * an f32 batch big enough to contain ties by itself is bit-identical to the f32 oracle, flags included.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _variants(N):
    from polardecoding_amd import testing as T
    if N == 1024:
        return [("auto", T.KERNEL_AUTO, "k_scl_fast"), ("two_per_wave", T.KERNEL_AUTO, "k_scl_fast"),
                ("four_per_wave", T.KERNEL_FOUR_PER_WAVE, "k_scl_fast4"), ("one_per_wave", T.KERNEL_ONE_PER_WAVE, "k_scl_fast<"),
                ("big", T.KERNEL_BIG, "k_scl_big"), ("generic", T.KERNEL_GENERIC, "k_scl_generic"),
                ("generic_spill", T.KERNEL_GENERIC_SPILL, "k_scl_generic")]
    return [("auto", T.KERNEL_AUTO, "k_scl_fast<"), ("generic", T.KERNEL_GENERIC, "k_scl_generic"),
            ("generic_spill", T.KERNEL_GENERIC_SPILL, "k_scl_generic")]


@pytest.mark.parametrize("name", ["SCL_128", "CASCL_128", "SCL_1024", "CASCL_1024_L8"])
def test_tie_fixtures_every_kernel(name, oracle):
    import polardecoding_amd as pa
    from polardecoding_amd import testing as T
    z = np.load(os.path.join(GOLD, f"ties_{name}.npz"))
    N, K, taps, algo, L = oracle.REF_PROGRAMS[name]
    sigma = float(z["sigma"])
    code = oracle.Code(N, K, taps)
    llr = np.stack([oracle.llr_from_y(y, sigma) for y in z["y"]])
    st = np.zeros((len(llr), 2), dtype=np.int32)
    o_uh, o_pm, o_ties = oracle.decode(code, llr, algo, L=L, stats=st)
    ret = z["returns"] == 1
    ref_tie = (z["diag"][:, 0] > 0) | ~ret
    assert ref_tie.sum() >= 8 and (st[:, 0] > 0).sum() > ref_tie.sum()
    for label, variant, expect in _variants(N):
        dec = pa.CASCL(N, K, L=L, crc_taps=taps) if algo == "CASCL" else pa.SCLdecode(N, K, L=L)
        T.select_kernel(dec, variant)
        assert expect in dec.kernel_name, (label, dec.kernel_name)
        # twice, the second time in reverse order: a frame's result must not depend on its neighbours in the batch
        for order in (np.arange(len(llr)), np.arange(len(llr))[::-1]):
            uh, pm, fl = dec.decode_batch_y(z["y"][order], sigma)
            inv = np.argsort(order)
            uh, pm, fl = uh[inv], pm[inv], fl[inv]
            assert np.array_equal(uh, o_uh) and np.array_equal(pm, o_pm), label
            assert np.array_equal((fl & pa.FLAG_TIE) != 0, ref_tie), label
            assert np.array_equal((fl & pa.FLAG_TIE) != 0, o_ties > 0), label
            assert np.array_equal(uh[ret], z["u_hat"][ret].astype(np.int32)) and np.array_equal(pm[ret], z["pm"][ret]), label
            if "generic" not in label:   # kernels that pre-rank on the metrics' high words
                assert np.array_equal((fl & pa.FLAG_RERANK) != 0, st[:, 0] > 0), label
            else:
                assert not (fl & pa.FLAG_RERANK).any()
        dec.close()


def _oracle_f32_parallel(oracle, code, llr32, algo, L, workers=12):
    """f32 oracle over a big batch on the box's host cores (ctypes releases the GIL)."""
    parts = np.array_split(np.arange(len(llr32)), workers * 4)
    def run(idx):
        return oracle.decode(code, llr32[idx], algo, L=L, dtype="f32")
    with ThreadPoolExecutor(workers) as ex:
        res = list(ex.map(run, parts))
    return (np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res]), np.concatenate([r[2] for r in res]))


@pytest.mark.parametrize("variant", ["auto", "four_per_wave", "one_per_wave", "big"])
def test_f32_batch_with_natural_ties_matches_f32_oracle(variant, oracle):
    """float metrics collide by themselves (0.1-0.25 % of frames at 1 dB): 24 576 frames of CA-SCL
    N = 1024 L = 8 at 1.0 dB, bit-identical to the f32 oracle: decisions, metric, and the tie flag."""
    import torch
    import polardecoding_amd as pa
    from polardecoding_amd import testing as T
    N, K, L, B = 1024, 512, 8, 24576
    dec = pa.CASCL(N, K, L=L, dtype=pa.F32)
    T.select_kernel(dec, dict(_v[:2] for _v in _variants(N))[variant])
    d_llr = torch.empty(B, N, dtype=torch.float32, device="cuda")
    dec.generate_device(20261004, 0, 1.0, d_llr)
    pm = torch.zeros(B, dtype=torch.float64, device="cuda")
    fl = torch.zeros(B, dtype=torch.int32, device="cuda")
    bits = dec.decode_device(d_llr, pm=pm, flags=fl)
    dec.synchronize()
    llr = d_llr.cpu().numpy()
    code = oracle.Code(N, K, pa.CRC24C_TAPS)
    o_uh, o_pm, o_ties = _oracle_f32_parallel(oracle, code, llr, "CASCL", L)
    w = bits.cpu().numpy().view(np.uint32)
    uh = ((w[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(B, N)
    flh = fl.cpu().numpy().view(np.uint32)
    assert (o_ties > 0).sum() >= 5, "the batch was meant to contain ties"
    assert np.array_equal(uh, o_uh)
    assert np.array_equal(pm.cpu().numpy().astype(np.float32), o_pm)
    assert np.array_equal((flh & pa.FLAG_TIE) != 0, o_ties > 0)
