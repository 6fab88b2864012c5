"""GPU: the reference's sequential stop rule `for (run = 0; errBlock < BLE; run++)` (SCL_1024.c:228, counters
:264-275) taken from the DEVICE's per-frame error counters (k_count_errors -> k_stop_cut), SURVEY 8(f2).
polar_sim's exact mode runs on this path, so tests/test_gpu_kat.py (every published run count) covers it end to end;
here the cut itself is checked against the host statement of the rule on constructed cases."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _host_cut(fe, need, min_frames=0):
    """main()'s loop, literally: frame after frame until `need` block errors have been seen (and, for the rule of the
    published L = 32 logs, at least min_frames frames)."""
    run = blk = bits = 0
    for e in fe:
        if blk >= need and run >= min_frames:
            break
        run += 1
        blk += int(e != 0)
        bits += int(e)
    return run, blk, bits


@pytest.mark.parametrize("B", [1, 63, 64, 1000, 1024, 1025, 5000, 70001])
def test_cut_kernel_against_the_loop(B):
    import torch
    import polardecoding_amd as pa
    dec = pa.SCLdecode(128, 64, L=8)
    rng = np.random.default_rng(B)
    out = torch.zeros(3, dtype=torch.int64, device="cuda")
    cases = []
    for p in (0.0, 0.002, 0.05, 0.5, 1.0):
        fe = (rng.random(B) < p) * rng.integers(1, 40, B)
        nbad = int((fe != 0).sum())
        for need in sorted({1, 2, max(1, nbad // 2), max(1, nbad), nbad + 1, 100000}):
            cases.append((fe.astype(np.uint32), need))
    # the erroneous frame exactly at the first / last position and at a 1024 boundary
    for pos in {0, B - 1, min(B - 1, 1023), min(B - 1, 1024)}:
        fe = np.zeros(B, dtype=np.uint32)
        fe[pos] = 7
        cases += [(fe, 1), (fe, 2)]
    for fe, need in cases:
        d = torch.from_numpy(fe.view(np.int32)).cuda()
        dec.stop_rule_cut_device(d, need, out)
        dec.synchronize()
        assert tuple(out.tolist()) == _host_cut(fe, need), (B, need)
        for mf in (1, B // 2 + 1, B, B + 5):      # `errBlock < BLE || run < min`
            for nd in (need, 0):
                dec.stop_rule_cut_device(d, nd, out, min_frames=mf)
                dec.synchronize()
                assert tuple(out.tolist()) == _host_cut(fe, nd, mf), (B, nd, mf)


@pytest.mark.parametrize("name", ["SCL_128", "CASCL_1024_L8"])
def test_stop_rule_batch_against_oracle(name, oracle):
    """decode + compare + cut on the device == the oracle decoding frame after frame under the same rule"""
    import polardecoding_amd as pa
    N, K, taps, algo, L = oracle.REF_PROGRAMS[name]
    code = oracle.Code(N, K, taps)
    dec = pa.CASCL(N, K, L=L, crc_taps=taps) if algo == "CASCL" else pa.SCLdecode(N, K, L=L)
    sig = oracle.sigma_from_db(1.0)
    B = 300 if N == 128 else 96
    us, ys = oracle.Sim(99).frames(code, sig, B)
    llr = np.stack([oracle.llr_from_y(y, sig) for y in ys])
    uh, _, _ = oracle.decode(code, llr, algo, L=L)
    fe = np.array([oracle.count_bit_errors(code, us[i], uh[i]) for i in range(B)])
    nbad = int((fe != 0).sum())
    assert nbad >= 5
    for need in (1, 3, nbad, nbad + 1):
        assert dec.stop_rule_batch_y(ys, sig, us, need) == _host_cut(fe, need), need
    assert dec.stop_rule_batch_y(ys, sig, us, 2, min_frames=B - 7) == _host_cut(fe, 2, B - 7)
