"""GPU: the work queue of the persistent kernels (csrc/polar_host.h work_queue(), csrc/polar_params.h next_job_*).

A launch with more jobs than resident wavefronts hands the jobs beyond the first round out through an atomic counter; which
wavefront decodes which frame then depends on timing.  Nothing about a frame's result may: one big launch must return,
frame for frame, what the same frames return in launches small enough to be assigned statically (the path the oracle and
golden-vector tests cover), again and again on the same context: the wavefront that takes the last number puts the counter
back to zero, so every launch -- and a replay of a captured one -- starts from the same state without a memset in between."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CRC24C = (0, 1, 2, 4, 8, 12, 13, 15, 17, 20, 21, 23, 24)
CRC6 = (0, 1, 6)

CASES = [
    # algo, N, K, kwargs, frames in the big launch (ragged on purpose), frames per small launch
    ("CASCL", 1024, 512, {"L": 8, "crc_taps": CRC24C}, 6144 * 5 + 1231, 4096),     # k_scl_fast2: 6144 frames resident
    ("SCL", 1024, 512, {"L": 8}, 6144 * 3 + 5, 4096),
    ("CASCL", 128, 64, {"L": 8, "crc_taps": CRC6}, 70001, 2048),                    # k_scl_fast
    ("SCL", 1024, 512, {"L": 32}, 4096 * 2 + 333, 2048),                            # k_scl_big, four wavefronts per SIMD
    ("CASCL", 2048, 1024, {"L": 32, "crc_taps": CRC24C}, 3072 * 2 + 77, 1024),      # k_scl_big with chain(), three per SIMD
    ("BP", 1024, 512, {"iterMax": 10}, 768 * 4 + 19, 512),                          # k_bp_r4: one job per workgroup
    ("BP", 128, 64, {"iterMax": 20}, 4096 * 6 + 3, 2048),                           # k_bp_w128
    ("SC", 1024, 512, {}, (1 << 18) + 777, 16384),                                  # k_sc_lanes: jobs are batches of 64
    ("SCL", 256, 128, {"L": 4}, 30011, 1024),                                       # k_scl_generic: one wavefront per workgroup
    ("BP", 256, 128, {"iterMax": 10}, 9001, 512),                                   # k_bp: rows in LDS, one job per workgroup
]


@pytest.mark.parametrize("algo,N,K,kw,B,chunk", CASES, ids=[f"{c[0]}_{c[1]}_{'L%d' % c[3]['L'] if 'L' in c[3] else 'x'}" for c in CASES])
def test_one_big_launch_equals_statically_assigned_small_launches(algo, N, K, kw, B, chunk):
    import torch
    import polardecoding_amd as pa
    from polardecoding_amd.synth import make_batch
    kw = dict(kw)
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
    gen.manual_seed(4242)
    llr, _ = make_batch(B, N, K, taps, 1.5, info, "cuda", gen)   # 1.5 dB: frames differ a lot in work (ranked steps, forks)
    lists = algo in ("CASCL", "SCL")

    def run(x):
        nb = x.shape[0]
        bits = torch.full((nb, N // 32), -1, dtype=torch.int32, device="cuda")
        pm = torch.full((nb,), -1.0, dtype=torch.float64, device="cuda") if lists else None
        fl = torch.full((nb,), -1, dtype=torch.int32, device="cuda") if lists else None
        dec.decode_device(x, out_bits=bits, pm=pm, flags=fl)
        return bits, pm, fl

    parts = [run(llr[i:i + chunk]) for i in range(0, B, chunk)]
    ref_bits = torch.cat([p[0] for p in parts])
    for rep in range(3):   # every launch must leave the counter at zero for the next one
        bits, pm, fl = run(llr)
        assert torch.equal(bits, ref_bits), f"decisions differ in launch {rep}"
        assert not bool((bits == -1).all(dim=1).any()), "a frame was not decoded"   # frozen positions are 0 in u_hat
        if lists:
            ref_pm = torch.cat([p[1] for p in parts])
            assert torch.equal(pm.view(torch.int64), ref_pm.view(torch.int64)), f"path metrics differ in launch {rep}"
            assert torch.equal(fl, torch.cat([p[2] for p in parts])), f"flags differ in launch {rep}"
        if rep == 0:
            run(llr[:7])   # a small launch in between (fixed stride, no queue) does not touch the counter


def test_captured_launch_can_be_replayed():
    """A launch captured in a HIP graph carries its kernel arguments with it: the queue must be back at zero after every
    replay without any host-side step in between (four replays of a launch of three rounds and a bit)."""
    import torch
    import polardecoding_amd as pa
    from polardecoding_amd.synth import make_batch
    dec = pa.CASCL(1024, 512, L=8, crc_taps=CRC24C)
    info = torch.tensor(dec.info_order.astype(np.int64), device="cuda")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    B = 6144 * 3 + 11
    llr, _ = make_batch(B, 1024, 512, CRC24C, 1.5, info, "cuda", gen)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        dec.use_torch_stream()
        ref = dec.decode_device(llr).clone()
        out = torch.full_like(ref, -1)
        dec.decode_device(llr, out_bits=out)   # every allocation is done before the capture
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            dec.use_torch_stream()
            dec.decode_device(llr, out_bits=out)
        for rep in range(4):
            out.fill_(-1)
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, ref), f"replay {rep}: {int((out == -1).all(dim=1).sum())} frames not decoded"


@pytest.mark.parametrize("algo,kw", [("CASCL", {"L": 8, "crc_taps": CRC24C}), ("BP", {"iterMax": 50})], ids=["CASCL_1024_L8", "BP_1024_50it"])
def test_frames_of_a_queued_launch_vs_oracle(algo, kw, oracle):
    """the headline configurations at a batch that goes through the queue (five rounds and a ragged rest): 64 frames from
    across the launch against the CPU oracle directly -- decisions, and for the list decoder the path metric, bit for bit"""
    import torch
    import polardecoding_amd as pa
    from polardecoding_amd.synth import make_batch
    kw = dict(kw)
    taps = kw.pop("crc_taps", None)
    dec = pa.CASCL(1024, 512, crc_taps=taps, **kw) if algo == "CASCL" else pa.BP(1024, 512, **kw)
    dec.use_torch_stream()
    info = torch.tensor(dec.info_order.astype(np.int64), device="cuda")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(99)
    B = 6144 * 5 + 1231 if algo == "CASCL" else 768 * 5 + 19
    llr, _ = make_batch(B, 1024, 512, taps, 1.5, info, "cuda", gen)
    pm = torch.zeros(B, dtype=torch.float64, device="cuda") if algo == "CASCL" else None
    bits = dec.decode_device(llr, pm=pm)
    torch.cuda.synchronize()
    rng = np.random.default_rng(1)
    pick = sorted(set([0, 1, B - 2, B - 1] + rng.integers(0, B, 60).tolist()))
    code = oracle.Code(1024, 512, taps)
    x = llr[pick].cpu().numpy()
    if algo == "CASCL":
        ref_uh, ref_pm, _ = oracle.decode(code, x, "CASCL", L=8)
    else:
        ref_uh = oracle.decode(code, x, "BP", bp_iters=50)[0]
    w = bits[pick].cpu().numpy().view(np.uint32)
    uh = ((w[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(len(pick), 1024)
    assert np.array_equal(uh, ref_uh)
    if algo == "CASCL":
        assert np.array_equal(pm[pick].cpu().numpy(), ref_pm)
