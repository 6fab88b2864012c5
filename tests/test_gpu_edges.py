"""GPU: edge cases of the boundary -- empty, single-frame, odd and ragged batches (the tuned kernel packs two
codewords per wavefront, so odd tails matter), frozen-mask override, y-input vs LLR-input equivalence,
error codes, context reuse, single-rank RCCL bench smoke."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, load_golden

pytestmark = pytest.mark.gpu


def _frames(oracle, N, K, taps, B, db, seed):
    code = oracle.Code(N, K, taps)
    sim = oracle.Sim(seed)
    sig = oracle.sigma_from_db(db)
    us, ys = sim.frames(code, sig, B)
    llr = np.stack([oracle.llr_from_y(y, sig) for y in ys])
    return code, sig, ys, llr


@pytest.mark.parametrize("B", [1, 2, 3, 5, 7, 17, 33])
def test_odd_and_small_batches_match_oracle(B, oracle):
    import polardecoding_amd as pa
    code, sig, ys, llr = _frames(oracle, 1024, 512, oracle.CRC24C_TAPS, B, 1.0, 100 + B)
    ref, ref_pm, _ = oracle.decode(code, llr, "CASCL", L=8)
    dec = pa.CASCL(1024, 512, L=8)
    uh, pm, fl = dec.decode_batch(llr)
    assert np.array_equal(uh, ref.reshape(B, -1))
    assert np.array_equal(pm, np.atleast_1d(ref_pm))


def test_empty_batch_is_a_noop():
    import polardecoding_amd as pa
    dec = pa.CASCL(1024, 512, L=8)
    uh, pm, fl = dec.decode_batch(np.zeros((0, 1024)))
    assert uh.shape == (0, 1024) and pm.shape == (0,)


def test_y_input_equals_llr_input(oracle):
    """polar_decode_batch_y forms 2*y/sigma/sigma inside the kernel: same bits as feeding those LLRs."""
    import polardecoding_amd as pa
    code, sig, ys, llr = _frames(oracle, 1024, 512, None, 9, 1.5, 5)
    dec = pa.SCLdecode(1024, 512, L=8)
    a, pa_, _ = dec.decode_batch_y(ys, sig)
    b, pb_, _ = dec.decode_batch(llr)
    assert np.array_equal(a, b) and np.array_equal(pa_, pb_)


def test_frozen_mask_override(oracle):
    """An explicit frozen mask (the reference's !inI[]) replaces the 5G set: SC and SCL, N = 128."""
    import polardecoding_amd as pa
    rng = np.random.default_rng(3)
    N, K = 128, 40
    info = np.sort(rng.choice(np.arange(20, N), size=K, replace=False))
    mask = np.ones(N, dtype=np.uint8)
    mask[info] = 0
    q = [j for j in range(N) if mask[j]] + info.tolist()
    code = oracle.Code(N, K, None, Q=q)
    sim = oracle.Sim(9)
    sig = oracle.sigma_from_db(3.0)
    us, ys = sim.frames(code, sig, 10)
    llr = np.stack([oracle.llr_from_y(y, sig) for y in ys])
    ref, _, _ = oracle.decode(code, llr, "SCL", L=8)
    dec = pa.SCLdecode(N, 64, L=8)              # built with the default set, overridden per call
    uh, _, _ = dec.decode_batch(llr, frozen_mask=mask)
    assert np.array_equal(uh, ref)
    assert np.array_equal(pa.decode(llr[0], mask, N, 8), ref[0])
    ref_sc, _, _ = oracle.decode(code, llr, "SC")
    assert np.array_equal(pa.decode(llr[0], mask, N, 1), ref_sc[0])


def test_context_reuse_and_interleaving(oracle):
    """Two contexts used alternately keep their own state (scratch, tables)."""
    import polardecoding_amd as pa
    g1, g2 = load_golden("CASCL_1024_L8"), load_golden("CASCL_128")
    d1, d2 = pa.CASCL(1024, 512, L=8), pa.CASCL(128, 64, L=8, crc_taps=pa.CRC6_TAPS)
    for i in range(4):
        assert np.array_equal(d1(g1["y"][i], float(g1["sigma"][i])), g1["u_hat"][i].astype(np.int32))
        assert np.array_equal(d2(g2["y"][i], float(g2["sigma"][i])), g2["u_hat"][i].astype(np.int32))


def test_errors_are_codes_not_crashes():
    import polardecoding_amd as pa
    dec = pa.CASCL(1024, 512, L=8)
    with pytest.raises(pa.PolarError):
        dec.decode_batch(np.zeros((2, 1024)), frozen_mask=np.zeros(1024, dtype=np.uint8))  # CASCL needs cfg's order
    with pytest.raises(pa.PolarError):
        dec.decode_batch_y(np.zeros((2, 1024)), 0.0)                                        # sigma must be > 0
    with pytest.raises(ValueError):
        dec(np.zeros(1000), 1.0)


def test_bench_under_torchrun_single_rank():
    """bench.py through torch.distributed.run with one rank: RCCL init, barrier, all-reduce path."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29577", os.path.join(REPO, "bench.py"),
                          "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "16384", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 1e5 and d["unit"] == "frames/s"
    for key in ("metric", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline"):
        assert key in d, key
    assert d["scaling"] == "weak" and d["dtype"] == "f64"
    # the path is VALU-issue bound: the HBM figures of the contract are reported, the binding roofline is named
    r = d["roofline"]
    assert r["bound"] == "valu" and r["unit"] == "GB/s" and 0 < r["frac"] < 0.05
    lo, hi = r["valu"]["achieved_frac"]
    assert 0.05 < lo < hi < 1.0 and r["valu"]["lane_ops_per_frame"] == [1267712, 1882112]
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["name"] == "cascl_1024_l8"
    assert d["rccl_ranks"] == 1 and len(d["devices"]) == 1 and d["devices"][0]   # what RCCL really summed over, and on what
    assert d["vs_baseline"] and d["vs_baseline"] > 1000
    # the drop-in's real cost (SURVEY 8d "kernel-only and end-to-end"): host buffers through polar_decode_batch, PCIe both ways,
    # and the latency of the literal per-frame replacement polar_decode(ctx, y, sigma, u_hat)
    e = d["end_to_end"]
    assert e["unit"] == "frames/s" and e["frames"] == 1 << 17 and 1e5 < e["value"] < d["single_launch_frames_per_s"]
    assert e["pcie_bytes"] == (1 << 17) * (1024 * 8 + 128) and 1.0 < e["pcie_GBps"] < 70.0
    lat = d["single_frame_latency_us"]
    assert lat["calls"] == 1000 and 20 < lat["p10"] <= lat["median"] <= lat["p90"] < 50000
    assert r["traffic_raw"] is None or r["traffic"] > r["traffic_raw"]   # reads corrected x 2 (only reported for the profiled batch)


def test_two_host_threads_two_contexts(oracle):
    """A ctx is not re-entrant, but two contexts on two host threads are independent (own stream, own scratch): both decode
    concurrently on the same GPU and both get the oracle's decisions; the calling thread's current device is left alone."""
    import threading
    import torch
    import polardecoding_amd as pa
    code = oracle.Code(1024, 512, oracle.CRC24C_TAPS)
    sig = oracle.sigma_from_db(1.5)
    _, ys = oracle.Sim(4).frames(code, sig, 40)
    llr = np.stack([oracle.llr_from_y(y, sig) for y in ys])
    ref, ref_pm, _ = oracle.decode(code, llr, "CASCL", L=8)
    big = np.tile(llr, (400, 1))                      # 16 000 frames per call: the two threads really overlap
    out = [None, None]
    dev_before = torch.cuda.current_device()

    def work(i):
        dec = pa.CASCL(1024, 512, L=8)
        for _ in range(3):
            uh, pm, _ = dec.decode_batch(big)
        out[i] = (uh, pm)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert torch.cuda.current_device() == dev_before
    for uh, pm in out:
        assert np.array_equal(uh.reshape(400, 40, 1024), np.broadcast_to(ref, (400, 40, 1024)))
        assert np.array_equal(pm.reshape(400, 40), np.broadcast_to(ref_pm, (400, 40)))


@pytest.mark.parametrize("in_dtype", ["f64", "f32"])
@pytest.mark.parametrize("shift", [0, 1])
def test_sc_channel_rows_read_in_place_any_alignment(in_dtype, shift, oracle):
    """k_sc_lanes reads the channel LLRs straight from the caller's device rows (16 elements per lane and burst, 16-byte
    vector loads when the buffer allows): a buffer that starts one element past a 16-byte boundary must take the scalar
    path and give the same decisions, for f64 and f32 input rows, with a ragged last batch of 64."""
    import torch
    import polardecoding_amd as pa
    N, K, B = 1024, 512, 64 * 3 + 11
    code = oracle.Code(N, K)
    sim = oracle.Sim(90210 + shift)
    sig = oracle.sigma_from_db(1.5)
    us, ys = sim.frames(code, sig, B)
    llr = np.stack([oracle.llr_from_y(y, sig) for y in ys]).astype(np.float32).astype(np.float64)
    ref_uh, _, _ = oracle.decode(code, llr, "SC")
    tdt = torch.float64 if in_dtype == "f64" else torch.float32
    flat = torch.empty(B * N + 4, dtype=tdt, device="cuda")
    x = flat[shift:shift + B * N].view(B, N)
    x.copy_(torch.from_numpy(llr).to(tdt))
    assert x.is_contiguous() and (x.data_ptr() % 16 == 0) == (shift == 0 or (in_dtype == "f32" and shift % 4 == 0))
    dec = pa.SCdecode(N, K)
    assert "k_sc_lanes" in dec.kernel_name
    bits = dec.decode_device(x)
    dec.synchronize()
    w = bits.cpu().numpy().view(np.uint32)
    uh = ((w[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(B, N).astype(np.int32)
    assert np.array_equal(uh, ref_uh)


@pytest.mark.parametrize("chunks", [3, 7], ids=["equal_chunks", "ramped_chunks"])
def test_host_batch_pipeline_equals_device_path(chunks):
    """polar_decode_batch on a batch of several chunks (pinned staging by helper threads, asynchronous DMA, decisions unpacked
    in the background) returns exactly what the device-pointer entry point returns on the same rows, into a caller-owned
    output array that is reused from call to call.  From six chunks on the chunk sizes ramp up and down (2048, 4096, 8192,
    16384 ... 16384, 8192, 4096, 2048 frames), with a ragged chunk in the middle."""
    import torch
    import polardecoding_amd as pa
    N, K, B = 128, 64, 16384 * chunks + 777
    rng = np.random.default_rng(4242)
    sigma = 10 ** (-1.5 / 20)
    llr = (2.0 * (1.0 + sigma * rng.standard_normal((B, N))) / sigma / sigma)
    dec = pa.CASCL(N, K, L=8, crc_taps=pa.CRC6_TAPS)
    out = np.full((B, N), -1, dtype=np.int32)
    uh, pm, fl = dec.decode_batch(llr, out=out)
    assert uh is out and set(np.unique(out).tolist()) <= {0, 1}
    x = torch.from_numpy(llr).cuda()
    pm_d = torch.empty(B, dtype=torch.float64, device="cuda")
    bits = dec.decode_device(x, pm=pm_d)
    dec.synchronize()
    w = bits.cpu().numpy().view(np.uint32)
    ref = ((w[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(B, N).astype(np.int32)
    assert np.array_equal(out, ref)
    assert np.array_equal(pm, pm_d.cpu().numpy())
    out.fill(-1)
    dec.decode_batch(llr[::-1].copy(), out=out)     # second call, same array, rows reversed
    assert np.array_equal(out, ref[::-1])
