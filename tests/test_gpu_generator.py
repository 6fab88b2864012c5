"""GPU: device-side transmit chain (polar_generate_device) -- encoder, CRC, noise statistics,
sharding invariance, and FER through generator + decoder against the reference's published BLER."""
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN, REPO

pytestmark = pytest.mark.gpu


def _unpack(words, N):
    w = words.cpu().numpy().view(np.uint32).reshape(-1, N // 32)
    return ((w[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(-1, N).astype(np.uint8)


def _polar(u):
    x = u.copy()
    s = 1
    N = x.shape[1]
    while s < N:
        v = x.reshape(x.shape[0], N // (2 * s), 2, s)
        v[:, :, 0, :] ^= v[:, :, 1, :]
        s *= 2
    return x


@pytest.mark.parametrize("N,K,taps", [(1024, 512, (0, 1, 2, 4, 8, 12, 13, 15, 17, 20, 21, 23, 24)), (128, 64, (0, 5, 6)),
                                      (1024, 512, None), (64, 32, None), (64, 24, (0, 5, 6)), (256, 100, (0, 5, 6))])
def test_generator_encoder_and_crc(N, K, taps):
    import torch
    import polardecoding_amd as pa
    dec = pa.CASCL(N, K, L=8, crc_taps=taps) if taps else pa.SCLdecode(N, K, L=8)
    B = 256
    GUARD = 4096                                              # the kernel must not write past frame B - 1
    ybuf = torch.full((B * N + GUARD,), 12345.0, dtype=torch.float64, device="cuda")
    y = ybuf[:B * N].view(B, N)
    ub = torch.empty(B, N // 32, dtype=torch.int32, device="cuda")
    dec.generate_device(7, 0, 40.0, y, ub, out_is_y=True)   # 40 dB: y = +-1 + 1e-2 noise
    dec.synchronize()
    assert bool((ybuf[B * N:] == 12345.0).all()), "generator wrote beyond its output"
    u = _unpack(ub, N)
    io = dec.info_order
    frozen = np.ones(N, bool)
    frozen[io] = False
    assert not u[:, frozen].any()
    assert 0.4 < u[:, io].mean() < 0.6
    x = _polar(u)
    hard = (y.cpu().numpy() < 0).astype(np.uint8)
    assert np.array_equal(hard, x)                            # BPSK of x = u F^{(x)n}
    if taps:                                                   # every frame is a multiple of g(D)
        r = max(taps)
        rem = u[:, io].copy()
        for i in range(rem.shape[1] - 1, r - 1, -1):
            rows = rem[:, i] == 1
            for t in taps:
                rem[rows, i - r + t] ^= 1
        assert not rem[:, :r].any()


def test_generator_systematic_crc_and_payload_metric():
    """crc_systematic = 1 (CASCL_1024_sys.c:776-789, :820-821): w[r..K+r) is the payload, w[0..r) the remainder that
    makes w(D) a multiple of g(D); the error counters ignore the r parity positions."""
    import torch
    import polardecoding_amd as pa
    taps = pa.CRC24C_TAPS
    r, N, K = max(taps), 1024, 512
    sysd = pa.CASCL(N, K, L=8, systematic=True)
    plain = pa.CASCL(N, K, L=8)
    B = 128
    y = torch.empty(B, N, dtype=torch.float64, device="cuda")
    ub = torch.empty(B, N // 32, dtype=torch.int32, device="cuda")
    ub2 = torch.empty(B, N // 32, dtype=torch.int32, device="cuda")
    sysd.generate_device(7, 0, 40.0, y, ub, out_is_y=True)
    plain.generate_device(7, 0, 40.0, y, ub2, out_is_y=True)
    sysd.synchronize(); plain.synchronize()
    io = sysd.info_order
    w = _unpack(ub, N)[:, io]
    w2 = _unpack(ub2, N)[:, io]
    # same Philox payload v in both modes: plain w2 = v g, so v is recovered by dividing; here simply re-encode
    v = w[:, r:]
    prod = np.zeros_like(w2)
    for t in taps:
        prod[:, t:t + K] ^= v
    assert np.array_equal(prod, w2)                 # the non-systematic word of the same payload
    rem = w.copy()
    for i in range(rem.shape[1] - 1, r - 1, -1):
        rows = rem[:, i] == 1
        for t in taps:
            rem[rows, i - r + t] ^= 1
    assert not rem[:, :r].any()                     # systematic word is a multiple of g(D)
    # error metric: flipping a parity position is not counted, flipping a payload position is
    uh = ub.clone()
    uh[:, io[0] // 32] ^= (1 << (int(io[0]) % 32)) if io[0] % 32 != 31 else -(1 << 31)
    cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
    sysd.count_errors_device(uh, ub, cnt)
    sysd.synchronize()
    assert cnt.tolist() == [0, 0]
    uh[:, io[r] // 32] ^= (1 << (int(io[r]) % 32)) if io[r] % 32 != 31 else -(1 << 31)
    cnt.zero_()
    sysd.count_errors_device(uh, ub, cnt)
    sysd.synchronize()
    assert cnt.tolist() == [B, B]


def test_crc_matrix_file_drives_the_product(tmp_path):
    """SURVEY 8f.3, file-format half: a context configured FROM the reference's generator-matrix file
    (polar_create_crc_file on tests/golden/CRC_6.dat: N = 128, K = 64, r = 6, systematic).  The parity bits the
    device-side transmit chain produces for 256 frames equal v * M over GF(2) with M as loaded from the file (what
    CASCL_1024_sys.c:776-789 does with its Gc literal); the context is the one CASCL(taps = {0,5,6}) gives; and
    `polar_sim --sys --crc-file CRC_6.dat` prints what `--sys --crc 6` prints."""
    import subprocess
    import torch
    import polardecoding_amd as pa
    from polardecoding_amd import crcfile
    path = os.path.join(GOLDEN, "CRC_6.dat")
    M = crcfile.load(path)                                   # [64][6], read by the independent Python reader
    N, K, r = 128, 64, 6
    dec = pa.CASCL(N, K, L=8, crc_file=path, systematic=True)
    ref = pa.CASCL(N, K, L=8, crc_taps=pa.CRC6_TAPS, systematic=True)
    assert dec.A == K + r and np.array_equal(dec.info_order, ref.info_order)
    B = 256
    y = torch.empty(B, N, dtype=torch.float64, device="cuda")
    ub = torch.empty(B, N // 32, dtype=torch.int32, device="cuda")
    dec.generate_device(11, 0, 2.0, y, ub, out_is_y=True)
    dec.synchronize()
    w = _unpack(ub, N)[:, dec.info_order]
    v = w[:, r:]
    assert v.any() and not v.all()
    assert np.array_equal(w[:, :r], (v.astype(np.int64) @ M.astype(np.int64)) & 1)
    # same frames, same decisions as the context built from the taps
    y2 = torch.empty_like(y); ub2 = torch.empty_like(ub)
    ref.generate_device(11, 0, 2.0, y2, ub2, out_is_y=True)
    ref.synchronize()
    assert torch.equal(y, y2) and torch.equal(ub, ub2)
    sig = 10 ** (-2.0 / 20)
    a = dec.decode_device(y, sigma=sig); dec.synchronize()
    b = ref.decode_device(y2, sigma=sig); ref.synchronize()
    assert torch.equal(a, b)
    # a file that is not a generator matrix does not make a context
    bad = M.copy(); bad[40, 3] ^= 1
    p = tmp_path / "bad.dat"
    p.write_bytes(crcfile.dumps(bad))
    with pytest.raises(pa.PolarError):
        pa.CASCL(N, K, L=8, crc_file=str(p), systematic=True)
    # the C harness: exact mode (host generator sums the file's rows like the reference sums Gc), fixed seed
    sim = os.path.join(REPO, "polardecoding_amd", "lib", "polar_sim")
    common = [sim, "--algo", "cascl", "--N", "128", "--K", "64", "--L", "8", "--sys", "--seed", "8392", "--ble", "50",
              "--snr", "1.0:2.0:0.5"]
    o1 = subprocess.run(common + ["--crc-file", path], capture_output=True, text=True, check=True).stdout
    o2 = subprocess.run(common + ["--crc", "6"], capture_output=True, text=True, check=True).stdout
    assert o1 == o2 and o1.count("BLER") == 3
    o3 = subprocess.run(common + ["--crc-file", str(p)], capture_output=True, text=True)
    assert o3.returncode != 0 and "not a CRC generator matrix" in o3.stderr


def test_generator_depends_only_on_seed_and_frame_index():
    import torch
    import polardecoding_amd as pa
    dec = pa.CASCL(1024, 512, L=8)
    a = torch.empty(64, 1024, dtype=torch.float64, device="cuda")
    b0 = torch.empty(24, 1024, dtype=torch.float64, device="cuda")
    b1 = torch.empty(40, 1024, dtype=torch.float64, device="cuda")
    ua = torch.empty(64, 32, dtype=torch.int32, device="cuda")
    dec.generate_device(99, 1000, 2.0, a, ua)
    dec.generate_device(99, 1000, 2.0, b0)
    dec.generate_device(99, 1024, 2.0, b1)
    dec.synchronize()
    assert torch.equal(a[:24], b0) and torch.equal(a[24:], b1)
    c = torch.empty(64, 1024, dtype=torch.float64, device="cuda")
    dec.generate_device(100, 1000, 2.0, c)
    dec.synchronize()
    assert not torch.equal(a, c)


def test_generator_noise_statistics():
    import torch
    import polardecoding_amd as pa
    dec = pa.CASCL(1024, 512, L=8)
    B = 1 << 13
    y = torch.empty(B, 1024, dtype=torch.float64, device="cuda")
    ub = torch.empty(B, 32, dtype=torch.int32, device="cuda")
    dec.generate_device(3, 0, 0.0, y, ub, out_is_y=True)      # sigma = 1
    dec.synchronize()
    x = torch.from_numpy(_polar(_unpack(ub, 1024))).cuda().double()
    nz = y - (1 - 2 * x)
    n = nz.numel()
    assert abs(nz.mean().item()) < 5 / math.sqrt(n)
    assert abs(nz.var().item() - 1) < 5 * math.sqrt(2 / n)
    assert abs((nz ** 4).mean().item() - 3) < 0.05
    # no correlation between neighbouring samples (the two Box-Muller outputs, neighbouring lanes)
    assert abs((nz[:, :-1] * nz[:, 1:]).mean().item()) < 5 / math.sqrt(n)
    assert abs((nz[:, :-64] * nz[:, 64:]).mean().item()) < 5 / math.sqrt(n)


def test_fer_with_device_generator_matches_published():
    """CA-SCL N=1024 L=8 at 1.5 dB: published BLER 0.0724 (CASCL_L8.dat, 100 errors)."""
    import torch
    import polardecoding_amd as pa
    dec = pa.CASCL(1024, 512, L=8)
    B = 1 << 15
    llr = torch.empty(B, 1024, dtype=torch.float64, device="cuda")
    ub = torch.empty(B, 32, dtype=torch.int32, device="cuda")
    dec.generate_device(2026, 0, 1.5, llr, ub)
    bits = dec.decode_device(llr)
    c = torch.zeros(2, dtype=torch.int64, device="cuda")
    dec.count_errors_device(bits, ub, c)
    dec.synchronize()
    fer = int(c[0]) / B
    ref = 0.072411
    sig = math.sqrt(ref * (1 - ref) / B + ref * ref / 100)
    assert abs(fer - ref) < 4 * sig, fer


def test_polar_sim_fast_mode_fer_curve():
    """The C harness in throughput mode (device generator + decode + count): CA-SCL N=1024 L=8 FER at
    1.0 / 1.5 / 2.0 / 2.5 dB against myResult_1024.zip:CASCL_L8.dat (SEED=1242 block, 100 errors each)."""
    import os
    import re
    import subprocess
    from conftest import REPO
    sim = os.path.join(REPO, "polardecoding_amd", "lib", "polar_sim")
    out = subprocess.run([sim, "--algo", "cascl", "--N", "1024", "--K", "512", "--L", "8", "--crc", "24c", "--seed", "11",
                          "--ble", "400", "--snr", "1.0:2.5:0.5", "--fast", "--batch", "65536"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-1000:]
    rows = re.findall(r"bSNR = ([\d.]+)\s+error block = (\d+)\s+run = (\d+)", out.stdout)
    ref = {1.0: 0.40650, 1.5: 0.072411, 2.0: 3.8414e-3, 2.5: 8.9455e-5}
    assert len(rows) == 4
    for snr, blk, run in rows:
        snr, blk, run = float(snr), int(blk), int(run)
        fer, r = blk / run, ref[snr]
        sig = math.sqrt(r * (1 - r) / run + r * r / 100)
        assert abs(fer - r) < 4 * sig, (snr, fer, r)


def test_fer_multi_gpu_entry_point_on_the_gpus_present():
    """polar_fer_multi_gpu (frames sharded over the node's GPUs, RCCL all-reduce of the two counters): with the GPUs this box
    has (one on the test box: RCCL initialises a one-rank communicator) the totals equal polar_fer_batch over the same
    frame range; asking for more GPUs than there are is an error code, not a crash."""
    import torch
    import polardecoding_amd as pa
    dec = pa.CASCL(1024, 512, L=8)
    n = min(torch.cuda.device_count(), 2)   # one on the test box; two ranks are enough to exercise the reduction elsewhere
    per = 8192
    want = dec.fer_batch(99, 1000, 1.5, per * n)
    blk, bits, sec = dec.fer_multi_gpu(n, 99, 1000, 1.5, per)
    assert (blk, bits) == want and blk > 100 and sec > 0
    with pytest.raises(pa.PolarError):
        dec.fer_multi_gpu(torch.cuda.device_count() + 1, 99, 1000, 1.5, per)
    # the C harness on the same entry points: --gpus 1 prints what the single-context path prints
    import os, subprocess
    sim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "polardecoding_amd", "lib", "polar_sim")
    args = [sim, "--algo", "cascl", "--N", "1024", "--K", "512", "--L", "8", "--crc", "24c", "--fast", "--batch", "32768",
            "--ble", "100", "--snr", "1.5:1.5:0.5", "--seed", "5"]
    a = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert a.returncode == 0 and "error block" in a.stdout


def test_group_exact_stop_rule_over_shards():
    """polar_group_stop_rule_batch: the reference's sequential stop rule (SCL_1024.c:228) over a batch decoded in shards --
    per-frame error counts gathered in frame order with ONE ncclAllGather, the cut on GPU 0.  With the GPUs present (one on
    the test box: a one-rank communicator, the collectives still run) the three numbers equal generate -> decode -> count
    -> polar_stop_rule_cut_device on a single context over the same frame range, for a cut inside the batch, for the
    minimum-run variant and for a batch that does not hold `need` errors.  polar_group_create's self-test (known values
    through all-reduce(uint64, sum) and all-gather(uint32)) has passed when the group exists."""
    import torch
    import polardecoding_amd as pa
    N, K = 1024, 512
    dec = pa.CASCL(N, K, L=8)
    n = min(torch.cuda.device_count(), 2)
    per = 6000                      # not a multiple of anything: ragged last wave, odd shard boundary
    total = per * n
    seed, first, snr = 4242, 17, 1.5
    llr = torch.empty(total, N, dtype=torch.float64, device="cuda")
    ub = torch.empty(total, N // 32, dtype=torch.int32, device="cuda")
    dec.generate_device(seed, first, snr, llr, ub)
    bits = dec.decode_device(llr)
    cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
    fe = torch.zeros(total, dtype=torch.int32, device="cuda")
    dec.count_errors_device(bits, ub, cnt, frame_err=fe)
    out = torch.zeros(3, dtype=torch.int64, device="cuda")
    grp = pa.Group(dec, n)
    assert grp.size == n
    nbad = int((fe != 0).sum().item())
    assert nbad > 60
    for need, min_frames in ((1, 0), (25, 0), (nbad, 0), (nbad + 5, 0), (3, 2000), (40, total)):
        dec.stop_rule_cut_device(fe, need, out, min_frames=min_frames)
        dec.synchronize()
        want = tuple(int(v) for v in out.tolist())
        got = grp.stop_rule_batch(seed, first, snr, per, need, min_frames=min_frames)
        assert got == want, (need, min_frames, got, want)
    # literal loop on the host for one of them
    fe_h = fe.cpu().numpy()
    run = blk = nbits = 0
    while blk < 25:
        blk += fe_h[run] != 0
        nbits += int(fe_h[run])
        run += 1
    assert grp.stop_rule_batch(seed, first, snr, per, 25) == (run, blk, nbits)
    # the plain counters of the same group agree with the single context too
    b2, e2, _ = grp.fer_batch(seed, first, snr, per)
    assert (b2, e2) == tuple(int(v) for v in cnt.tolist())
    grp.close()
