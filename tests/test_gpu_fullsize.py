"""GPU, BASELINE.json's full batch sizes: size-independent properties of the decode path (the oracle cannot
finish 2^16..2^17 frames in seconds): noiseless / high-SNR round trips through encode -> channel -> decode,
determinism, batch-permutation equivariance, CRC flag consistency, and FER against the reference's
published block-error rates within Monte-Carlo error."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PUBLISHED_CASCL_1024_L8 = {1.0: 0.40650, 1.5: 0.072411, 2.0: 3.8414e-3}   # myResult_1024.zip:CASCL_L8.dat, SEED=1242
PUBLISHED_SCL_1024_L8 = {1.0: 0.22026, 1.5: 0.048733, 2.0: 8.5222e-3}      # SCL1024out.dat, L = 8 block
PUBLISHED_BP_1024 = {1.5: 0.1546, 2.0: 0.032916, 2.5: 5.675e-3}            # BP1024out_NewSEED.dat, SEED=771 (100 iters)


def _setup(algo, N, K, B, snr, seed=11, **kw):
    import torch
    import polardecoding_amd as pa
    from polardecoding_amd.synth import make_batch
    taps = kw.pop("crc_taps", None)
    if algo == "CASCL":
        dec = pa.CASCL(N, K, crc_taps=taps, **kw)
    elif algo == "SCL":
        dec = pa.SCLdecode(N, K, **kw)
    elif algo == "BP":
        dec = pa.BP(N, K, **kw)
    else:
        dec = pa.SCdecode(N, K, **kw)
    dec.use_torch_stream()
    info = torch.tensor(dec.info_order.astype(np.int64), device="cuda")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    llr, u_words = make_batch(B, N, K, taps, snr, info, "cuda", gen)
    return dec, llr, u_words


def _errors(dec, bits, u_words):
    import torch
    c = torch.zeros(2, dtype=torch.int64, device="cuda")
    fe = torch.zeros(bits.shape[0], dtype=torch.int32, device="cuda")
    dec.count_errors_device(bits, u_words, c, fe)
    torch.cuda.synchronize()
    return int(c[0]), int(c[1]), fe


@pytest.mark.parametrize("algo,B,kw", [("CASCL", 1 << 17, {"L": 8, "crc_taps": (0, 1, 2, 4, 8, 12, 13, 15, 17, 20, 21, 23, 24)}),
                                       ("SCL", 1 << 16, {"L": 8}), ("BP", 1 << 16, {"iterMax": 50}), ("SC", 1 << 16, {})])
def test_high_snr_round_trip_full_batch(algo, B, kw):
    """encode -> BPSK -> AWGN at 8 dB -> decode returns exactly what was sent, for the whole batch."""
    dec, llr, u_words = _setup(algo, 1024, 512, B, 8.0, **kw)
    bits = dec.decode_device(llr)
    blk, nbits, _ = _errors(dec, bits, u_words)
    assert (blk, nbits) == (0, 0)


def test_deterministic_and_permutation_equivariant():
    import torch
    dec, llr, u_words = _setup("CASCL", 1024, 512, 1 << 16, 1.5, L=8,
                               crc_taps=(0, 1, 2, 4, 8, 12, 13, 15, 17, 20, 21, 23, 24))
    a = dec.decode_device(llr).clone()
    b = dec.decode_device(llr).clone()
    assert torch.equal(a, b)
    perm = torch.randperm(llr.shape[0], device="cuda")
    c = dec.decode_device(llr[perm].contiguous())
    assert torch.equal(c, a[perm])


def test_crc_flag_consistent_with_output():
    """POLAR_FLAG_CRC_PASS <=> the returned word satisfies the CRC (checked by re-dividing on the host)."""
    import torch
    import polardecoding_amd as pa
    taps = pa.CRC24C_TAPS
    dec, llr, u_words = _setup("CASCL", 1024, 512, 1 << 14, 1.0, L=8, crc_taps=taps)
    flags = torch.zeros(llr.shape[0], dtype=torch.int32, device="cuda")
    bits = dec.decode_device(llr, flags=flags)
    torch.cuda.synchronize()
    w = bits.cpu().numpy().view(np.uint32)
    uh = ((w[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(-1, 1024)
    io = dec.info_order
    cw = uh[:, io].astype(np.uint8)          # C[i] = u_hat[I[i]]
    rem = cw.copy()
    r = 24
    for i in range(cw.shape[1] - 1, r - 1, -1):
        rows = rem[:, i] == 1
        for t in taps:
            rem[rows, i - r + t] ^= 1
    ok = ~rem[:, :r].any(axis=1)
    fl = (flags.cpu().numpy() & pa.FLAG_CRC_PASS) != 0
    assert np.array_equal(ok, fl)
    assert 0.3 < ok.mean() < 0.9            # at 1 dB roughly 60 % of the frames end on a CRC-passing path


def _fer_check(algo, published, frames_at, **kw):
    for snr, ref in published.items():
        B = frames_at[snr]
        dec, llr, u_words = _setup(algo, 1024, 512, B, snr, seed=int(snr * 100), **kw)
        bits = dec.decode_device(llr)
        blk, _, _ = _errors(dec, bits, u_words)
        fer = blk / B
        # Monte-Carlo error: ours (binomial) and the published point's (100-200 errors), 4 sigma
        sig = math.sqrt(ref * (1 - ref) / B + ref * ref / 100.0)
        assert abs(fer - ref) < 4 * sig, f"{algo} @ {snr} dB: FER {fer:.4g} vs published {ref:.4g}"


def test_fer_overlays_published_cascl():
    _fer_check("CASCL", PUBLISHED_CASCL_1024_L8, {1.0: 1 << 13, 1.5: 1 << 15, 2.0: 1 << 17}, L=8,
               crc_taps=(0, 1, 2, 4, 8, 12, 13, 15, 17, 20, 21, 23, 24))


def test_fer_overlays_published_scl():
    _fer_check("SCL", PUBLISHED_SCL_1024_L8, {1.0: 1 << 13, 1.5: 1 << 15, 2.0: 1 << 16}, L=8)


def test_fer_overlays_published_bp():
    _fer_check("BP", PUBLISHED_BP_1024, {1.5: 1 << 13, 2.0: 1 << 15, 2.5: 1 << 16}, iterMax=100)


def test_f32_fer_matches_f64_and_mismatch_rate_is_small():
    """f32 arithmetic: same FER, a few frames in 10^4 decided differently."""
    import torch
    import polardecoding_amd as pa
    taps = pa.CRC24C_TAPS
    dec64, llr, u_words = _setup("CASCL", 1024, 512, 1 << 15, 1.0, L=8, crc_taps=taps)
    dec32 = pa.CASCL(1024, 512, L=8, crc_taps=taps, dtype=pa.F32)
    dec32.use_torch_stream()
    a = dec64.decode_device(llr)
    b = dec32.decode_device(llr.float().contiguous())
    torch.cuda.synchronize()
    differ = int((a != b).any(dim=1).sum())
    e64, _, _ = _errors(dec64, a, u_words)
    e32, _, _ = _errors(dec64, b, u_words)
    assert differ < 0.01 * llr.shape[0]
    assert abs(e64 - e32) <= max(20, 0.02 * e64)


# ---- BASELINE config 5 at its real per-GPU batch: N = 4096, K = 2048, CA-SCL L = 32, 2^15 frames (2^18 over 8 GPUs) ----
# No reference exists above N = 1024 (parity unpinned): what can be shown is self-consistency with the
# oracle on frames drawn from ACROSS the big batch -- i.e. from every region of the 5120-codeword-wide scratch
# slicing (5.6 GB, "LLRs spill HBM") that a 6-frame test never reaches -- plus the size-independent properties.

def _config5(B, snr, seed):
    import polardecoding_amd as pa
    return _setup("CASCL", 4096, 2048, B, snr, seed=seed, L=32, crc_taps=pa.CRC24C_TAPS)


def test_config5_full_batch_round_trip_and_determinism():
    import torch
    B = 1 << 15
    dec, llr, u_words = _config5(B, 8.0, 5)
    assert "k_scl_big" in dec.kernel_name
    bits = dec.decode_device(llr)
    blk, nbits, _ = _errors(dec, bits, u_words)
    assert (blk, nbits) == (0, 0)                       # 8 dB: every frame comes back as sent
    del llr, u_words, bits
    dec, llr, u_words = _config5(B, 1.5, 6)
    a = dec.decode_device(llr).clone()
    b = dec.decode_device(llr).clone()
    assert torch.equal(a, b)                            # same launch twice
    perm = torch.randperm(B, device="cuda")
    c = dec.decode_device(llr[perm].contiguous())
    assert torch.equal(c, a[perm])                      # a frame's result does not depend on its place in the batch


def test_config5_full_batch_samples_vs_oracle(oracle):
    """eight frames taken from across a 2^15-frame launch (first and last workgroups, both ends of the resident set,
    the ragged tail) against the oracle: decisions and path metric"""
    import torch
    import polardecoding_amd as pa
    from test_gpu_parity import _oracle_code_like
    B = 1 << 15
    dec, llr, u_words = _config5(B, 1.5, 7)
    pm = torch.zeros(B, dtype=torch.float64, device="cuda")
    bits = dec.decode_device(llr, pm=pm)
    torch.cuda.synchronize()
    code = _oracle_code_like(oracle, dec, 4096, 2048, pa.CRC24C_TAPS)
    pick = [0, 1, 5119, 5120, 12345, 20479, B - 2, B - 1]
    ref_uh, ref_pm, _ = oracle.decode(code, llr[pick].cpu().numpy(), "CASCL", L=32)
    w = bits[pick].cpu().numpy().view(np.uint32)
    uh = ((w[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(len(pick), 4096)
    assert np.array_equal(uh, ref_uh)
    assert np.array_equal(pm[pick].cpu().numpy(), ref_pm)
