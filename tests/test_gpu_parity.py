"""GPU parity tests proper: the HIP path (through the C ABI) against the committed golden vectors
(generated from the compiled reference) and against the CPU oracle on seeded inputs.  Bit-exact in f64."""
import numpy as np
import pytest

from conftest import load_golden, unpack_bits

pytestmark = pytest.mark.gpu

PROGRAMS = {
    # name: (N, K, algo ctor, kwargs)
    "SC_128": (128, 64, "SCdecode", {}),
    "SC_1024": (1024, 512, "SCdecode", {}),
    "BP_128": (128, 64, "BP", {"iterMax": 100}),
    "BP_1024": (1024, 512, "BP", {"iterMax": 100}),
    "BP_1024_it50": (1024, 512, "BP", {"iterMax": 50}),   # BASELINE config 2's count; fixture from BP_1024.c compiled with iterMax 50
    "SCL_128": (128, 64, "SCLdecode", {"L": 8}),
    "SCL_1024": (1024, 512, "SCLdecode", {"L": 8}),
    "CASCL_128": (128, 64, "CASCL", {"L": 8, "crc_taps": (0, 5, 6)}),
    "CASCL_1024_L8": (1024, 512, "CASCL", {"L": 8}),
    # CASCL_1024_sys.c decodes on the bit-reversed graph with y[bRev[j]] on channel row j: the same decisions as the
    # natural-order decoder on y, which these fixtures (made by the compiled program) show frame by frame
    "CASCL_1024_sys": (1024, 512, "CASCL", {"L": 8, "systematic": True}),
}


def make(name, **extra):
    import polardecoding_amd as pa
    N, K, ctor, kw = PROGRAMS[name]
    kw = dict(kw)
    kw.update(extra)
    return getattr(pa, ctor)(N, K, **kw)


@pytest.mark.parametrize("name", list(PROGRAMS))
def test_golden_reference_shape(name):
    """polar_decode(y, sigma, u_hat) == reference X(y, u_hat) on every golden frame."""
    g = load_golden(name)
    dec = make(name)
    for i in range(min(12, len(g["sigma"]))):
        uh = dec(g["y"][i], float(g["sigma"][i]))
        assert np.array_equal(uh, g["u_hat"][i].astype(np.int32)), f"{name} frame {i}"


@pytest.mark.parametrize("name", list(PROGRAMS))
def test_golden_batch_y(name):
    """Batched entry with in-kernel 2*y/sigma/sigma: decisions and path metric bit-identical."""
    g = load_golden(name)
    dec = make(name)
    for s in np.unique(g["sigma"]):
        sel = g["sigma"] == s
        uh, pm, fl = dec.decode_batch_y(g["y"][sel], float(s))
        assert np.array_equal(uh, g["u_hat"][sel].astype(np.int32)), name
        if name.startswith(("SCL", "CASCL")):
            assert np.array_equal(pm, g["pm"][sel]), name
            assert not (fl & 1).any()


@pytest.mark.parametrize("name", list(PROGRAMS))
def test_golden_batch_llr(name, oracle):
    g = load_golden(name)
    dec = make(name)
    llr = np.stack([oracle.llr_from_y(y, float(s)) for y, s in zip(g["y"], g["sigma"])])
    uh, pm, fl = dec.decode_batch(llr)
    assert np.array_equal(uh, g["u_hat"].astype(np.int32))


@pytest.mark.parametrize("L", [1, 2, 4, 16, 32])
@pytest.mark.parametrize("N,K", [(128, 64), (1024, 512)])
def test_scl_list_sizes_vs_oracle(N, K, L, oracle):
    """List sizes the reference's logs cover (SCL1024out.dat: L = 2..32) against the oracle."""
    import polardecoding_amd as pa
    code = oracle.Code(N, K)
    sim = oracle.Sim(1234 + L)
    B = 24 if N == 1024 else 64
    sig = oracle.sigma_from_db(1.5)
    us, ys = sim.frames(code, sig, B)
    llr = np.stack([oracle.llr_from_y(y, sig) for y in ys])
    ref_uh, ref_pm, _ = oracle.decode(code, llr, "SCL", L=L)
    dec = pa.SCLdecode(N, K, L=L)
    uh, pm, fl = dec.decode_batch(llr)
    assert np.array_equal(uh, ref_uh)
    assert np.array_equal(pm, ref_pm)


def _oracle_code_like(oracle, dec, N, K, taps):
    """oracle Code with exactly the decoder's information order (needed above N = 1024, where the reference
    has no reliability table and the library builds the beta-expansion order)."""
    io = dec.info_order
    rest = [j for j in range(N) if j not in set(io.tolist())]
    return oracle.Code(N, K, taps, Q=rest + io.tolist())


@pytest.mark.parametrize("N,K,L,crc", [(4096, 2048, 32, True), (4096, 2048, 8, False), (2048, 1024, 16, True)])
def test_config5_spilled_levels_vs_oracle(N, K, L, crc, oracle):
    """BASELINE config 5 (N=4096 K=2048 CA-SCL L=32; no counterpart in the reference: parity unpinned,
    oracle <-> GPU self-consistency only).  LLR levels live in global scratch ("LLRs spill HBM")."""
    import polardecoding_amd as pa
    taps = pa.CRC24C_TAPS if crc else None
    dec = pa.CASCL(N, K, L=L, crc_taps=taps) if crc else pa.SCLdecode(N, K, L=L)
    code = _oracle_code_like(oracle, dec, N, K, taps)
    sim = oracle.Sim(31 + L)
    sig = oracle.sigma_from_db(1.5)
    B = 6
    us, ys = sim.frames(code, sig, B)
    llr = np.stack([oracle.llr_from_y(y, sig) for y in ys])
    ref_uh, ref_pm, _ = oracle.decode(code, llr, "CASCL" if crc else "SCL", L=L)
    uh, pm, fl = dec.decode_batch(llr)
    assert np.array_equal(uh, ref_uh)
    assert np.array_equal(pm, ref_pm)


def test_config5_f32_kernel_vs_f32_oracle(oracle):
    """the f32 instantiation of the config-5 kernel (chain(), split 4 / 7 / 1, three wavefronts per SIMD) keeps the operation
    order: bit-identical to the oracle's f32 instantiation (decisions; the path metric as float)"""
    import polardecoding_amd as pa
    N, K, L = 4096, 2048, 32
    dec = pa.CASCL(N, K, L=L, crc_taps=pa.CRC24C_TAPS, dtype=pa.F32)
    code = _oracle_code_like(oracle, dec, N, K, pa.CRC24C_TAPS)
    sim = oracle.Sim(977)
    sig = oracle.sigma_from_db(1.5)
    us, ys = sim.frames(code, sig, 6)
    llr = np.stack([oracle.llr_from_y(y, sig) for y in ys]).astype(np.float32).astype(np.float64)
    ref_uh, ref_pm, _ = oracle.decode(code, llr, "CASCL", L=L, dtype="f32")
    uh, pm, fl = dec.decode_batch(llr)
    assert np.array_equal(uh, ref_uh)
    assert np.array_equal(np.asarray(pm, dtype=np.float32), np.asarray(ref_pm, dtype=np.float32))


def test_forced_spill_matches_golden():
    """The global-scratch variant on a shape that also fits LDS: same bits as the reference."""
    import polardecoding_amd as pa
    from polardecoding_amd import testing as T
    g = load_golden("CASCL_1024_L8")
    dec = T.select_kernel(pa.CASCL(1024, 512, L=8), T.KERNEL_GENERIC_SPILL)
    assert "generic" in dec.kernel_name
    for s in np.unique(g["sigma"]):
        sel = g["sigma"] == s
        uh, pm, fl = dec.decode_batch_y(g["y"][sel], float(s))
        assert np.array_equal(uh, g["u_hat"][sel].astype(np.int32))
        assert np.array_equal(pm, g["pm"][sel])


@pytest.mark.parametrize("N,K,B", [(1024, 512, 200), (128, 64, 333), (64, 20, 65), (32, 16, 130), (512, 400, 64), (1024, 40, 70),
                                   (2048, 1024, 66)])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_sc_one_codeword_per_lane_vs_oracle(N, K, B, dtype, oracle):
    """sc_lanes.h (B >= 64): 64 codewords per wavefront, frozen subtrees skipped -- same decisions as the oracle
    (= SC_128.c / SC_1024.c on the fixtures), ragged last batch, short and long codes, low and high rate."""
    import polardecoding_amd as pa
    dec = pa.SCdecode(N, K, dtype=pa.F64 if dtype == "f64" else pa.F32)
    code = oracle.Code(N, K) if N <= 1024 else _oracle_code_like(oracle, dec, N, K, None)
    sim = oracle.Sim(555 + N + K)
    sig = oracle.sigma_from_db(1.0 if K * 2 <= N else 3.0)
    us, ys = sim.frames(code, sig, B)
    llr = np.stack([oracle.llr_from_y(y, sig) for y in ys]).astype(np.float32).astype(np.float64)
    ref_uh, _, _ = oracle.decode(code, llr, "SC", dtype=dtype)
    uh, pm, fl = dec.decode_batch(llr)
    assert np.array_equal(uh, ref_uh)
    # the y-input form (LLR formed in the kernel) on the same frames
    if dtype == "f64":
        uh2, _, _ = dec.decode_batch_y(ys, sig)
        ref2, _, _ = oracle.decode(code, np.stack([oracle.llr_from_y(y, sig) for y in ys]), "SC")
        assert np.array_equal(uh2, ref2)


def test_big_list_kernel_matches_golden():
    """scl_big.h (low levels in LDS, the rest in scratch, lazily shared partial sums) forced onto the
    reference's own N = 1024 L = 8 programs: same bits and path metrics as the compiled reference."""
    from polardecoding_amd import testing as T
    for name in ("CASCL_1024_L8", "SCL_1024"):
        g = load_golden(name)
        dec = T.select_kernel(make(name), T.KERNEL_BIG)
        assert "k_scl_big" in dec.kernel_name
        for s in np.unique(g["sigma"]):
            sel = g["sigma"] == s
            uh, pm, fl = dec.decode_batch_y(g["y"][sel], float(s))
            assert np.array_equal(uh, g["u_hat"][sel].astype(np.int32))
            assert np.array_equal(pm, g["pm"][sel])


@pytest.mark.parametrize("L", [2, 4, 16, 32])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_big_list_kernel_list_sizes_vs_oracle(L, dtype, oracle):
    import polardecoding_amd as pa
    N, K = 1024, 512
    code = oracle.Code(N, K, pa.CRC24C_TAPS)
    sim = oracle.Sim(77 + L)
    sig = oracle.sigma_from_db(1.0)
    us, ys = sim.frames(code, sig, 24)
    llr = np.stack([oracle.llr_from_y(y, sig) for y in ys]).astype(np.float32).astype(np.float64)
    ref_uh, ref_pm, _ = oracle.decode(code, llr, "CASCL", L=L, dtype=dtype)
    dec = pa.CASCL(N, K, L=L, dtype=pa.F64 if dtype == "f64" else pa.F32)
    assert "k_scl_big" in dec.kernel_name
    uh, pm, fl = dec.decode_batch(llr)
    assert np.array_equal(uh, ref_uh)
    if dtype == "f64":
        assert np.array_equal(pm, ref_pm)


@pytest.mark.parametrize("split", [35, 46, 57, 351, 371])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_big_list_kernel_storage_splits_vs_oracle(split, dtype, oracle):
    """Every LDS / register / scratch split of k_scl_big (polar_testing_big_split), L = 32: decisions, path metric and
    flags of the oracle (SCL_1024.c:547-680 with 32 paths), whichever split the library would pick by itself."""
    import polardecoding_amd as pa
    from polardecoding_amd import testing as T
    N, K, L = 1024, 512, 32
    code = oracle.Code(N, K, pa.CRC24C_TAPS)
    sim = oracle.Sim(4100 + split)
    sig = oracle.sigma_from_db(1.0)
    us, ys = sim.frames(code, sig, 40)
    llr = np.stack([oracle.llr_from_y(y, sig) for y in ys]).astype(np.float32).astype(np.float64)
    ref_uh, ref_pm, ref_ties = oracle.decode(code, llr, "CASCL", L=L, dtype=dtype)
    dec = T.big_split(pa.CASCL(N, K, L=L, dtype=pa.F64 if dtype == "f64" else pa.F32), split)
    uh, pm, fl = dec.decode_batch(llr)
    assert np.array_equal(uh, ref_uh)
    assert np.array_equal((fl & pa.FLAG_TIE) != 0, ref_ties > 0)
    if dtype == "f64":
        assert np.array_equal(pm, ref_pm)
    else:
        assert np.array_equal(np.asarray(pm, dtype=np.float32), np.asarray(ref_pm, dtype=np.float32))


@pytest.mark.parametrize("name", ["SC_1024", "SCL_1024", "CASCL_1024_L8", "CASCL_128", "BP_128"])
def test_f32_matches_f32_oracle(name, oracle):
    """The f32 kernels keep the operation order: bit-identical to the oracle's f32 instantiation."""
    import polardecoding_amd as pa
    N, K, ctor, kw = PROGRAMS[name]
    g = load_golden(name)
    code = oracle.Code(N, K, kw.get("crc_taps", pa.CRC24C_TAPS) if ctor == "CASCL" else None)
    llr = np.stack([oracle.llr_from_y(y, float(s)) for y, s in zip(g["y"], g["sigma"])])
    llr = llr.astype(np.float32).astype(np.float64)  # exactly representable inputs for both sides
    algo = {"SCdecode": "SC", "BP": "BP", "SCLdecode": "SCL", "CASCL": "CASCL"}[ctor]
    ref_uh, _, _ = oracle.decode(code, llr, algo, L=kw.get("L", 1), bp_iters=100, dtype="f32")
    dec = make(name, dtype=pa.F32)
    uh, pm, fl = dec.decode_batch(llr)
    assert np.array_equal(uh, ref_uh)


def test_north_star_call_shape(oracle):
    """decode(llr_in, frozen_mask, N, L)"""
    import polardecoding_amd as pa
    code = oracle.Code(1024, 512)
    sim = oracle.Sim(5)
    sig = oracle.sigma_from_db(2.0)
    u, y = sim.frame(code, sig)
    llr = oracle.llr_from_y(y, sig)
    ref, _, _ = oracle.decode(code, llr, "SCL", L=8)
    out = pa.decode(llr, code.frozen, 1024, 8)
    assert np.array_equal(out, ref)


def test_device_path_and_error_count(oracle):
    """Device-pointer entry + device error accounting vs the oracle's counts."""
    import torch
    import polardecoding_amd as pa
    code = oracle.Code(1024, 512, oracle.CRC24C_TAPS)
    sim = oracle.Sim(99)
    sig = oracle.sigma_from_db(1.0)
    B = 48
    us, ys = sim.frames(code, sig, B)
    llr = np.stack([oracle.llr_from_y(y, sig) for y in ys])
    ref_uh, _, _ = oracle.decode(code, llr, "CASCL", L=8)
    dec = pa.CASCL(1024, 512, L=8)
    dec.use_torch_stream()
    d_llr = torch.from_numpy(llr).cuda()
    bits = dec.decode_device(d_llr)
    torch.cuda.synchronize()
    uh = unpack_bits(bits.cpu().numpy(), 1024)
    assert np.array_equal(uh, ref_uh)
    # error accounting
    u_words = np.zeros((B, 32), dtype=np.uint32)
    for b in range(B):
        for j in np.nonzero(us[b])[0]:
            u_words[b, j >> 5] |= np.uint32(1) << np.uint32(j & 31)
    d_u = torch.from_numpy(u_words.view(np.int32)).cuda()
    counters = torch.zeros(2, dtype=torch.int64, device="cuda")
    ferr = torch.zeros(B, dtype=torch.int32, device="cuda")
    dec.count_errors_device(bits, d_u, counters, ferr)
    torch.cuda.synchronize()
    exp = np.array([oracle.count_bit_errors(code, us[b], ref_uh[b]) for b in range(B)])
    assert np.array_equal(ferr.cpu().numpy(), exp)
    assert counters.cpu().tolist() == [int((exp > 0).sum()), int(exp.sum())]


def _pack_bits(u):
    u = np.asarray(u, dtype=np.uint32)
    B, N = u.shape
    w = (u.reshape(B, N // 32, 32) << np.arange(32, dtype=np.uint32)).sum(axis=2, dtype=np.uint64).astype(np.uint32)
    return w.view(np.int32)


def test_bp_readouts_match_compiled_bpr_fixture():
    """BPr_128.c: polar_bp_readout_device reproduces the compiled program's decisions and its per-stage
    read-out table E[6][n+1] on every fixture frame (iterMax 90, read-outs after 3, 6, 10, 20, 40, 80 iterations)."""
    import torch
    import polardecoding_amd as pa
    g = load_golden("BPr_128")
    iters, cp = int(g["iters"]), g["checkpoints"].tolist()
    dec = pa.BP(128, 64, iterMax=iters)
    F = len(g["sigma"])
    ub = torch.from_numpy(_pack_bits(g["u"])).cuda()
    for i in range(F):   # per frame: the fixture holds E frame by frame
        y = torch.from_numpy(g["y"][i:i + 1].copy()).cuda()
        E = torch.zeros(len(cp), 8, dtype=torch.int64, device="cuda")
        out = torch.zeros(1, 4, dtype=torch.int32, device="cuda")
        dec.bp_readout_device(y, ub[i:i + 1].contiguous(), cp, E, out_bits=out, sigma=float(g["sigma"][i]))
        dec.synchronize()
        assert np.array_equal(E.cpu().numpy(), g["E"][i]), i
        assert np.array_equal(unpack_bits(out.cpu().numpy(), 128)[0], g["u_hat"][i]), i
    # one launch over the frames of one noise level: the table is the sum
    sel = np.nonzero(g["sigma"] == g["sigma"][0])[0]
    y = torch.from_numpy(g["y"][sel].copy()).cuda()
    E = torch.zeros(len(cp), 8, dtype=torch.int64, device="cuda")
    dec.bp_readout_device(y, ub[sel].contiguous(), cp, E, sigma=float(g["sigma"][0]))
    dec.synchronize()
    assert np.array_equal(E.cpu().numpy(), g["E"][sel].sum(axis=0))


def test_bp_readouts_vs_oracle_n512(oracle):
    import torch
    import polardecoding_amd as pa
    N, K, iters, cp = 512, 256, 30, [1, 5, 30]
    code = oracle.Code(N, K)
    sim = oracle.Sim(31337)
    sig = oracle.sigma_from_db(2.0)
    us, ys = sim.frames(code, sig, 12)
    llr = np.stack([oracle.llr_from_y(y, sig) for y in ys])
    ref_uh, ref_E = oracle.bp_readout(code, llr, us, iters, cp)
    dec = pa.BP(N, K, iterMax=iters)
    E = torch.zeros(len(cp), code.n + 1, dtype=torch.int64, device="cuda")
    out = torch.zeros(12, N // 32, dtype=torch.int32, device="cuda")
    dec.bp_readout_device(torch.from_numpy(llr).cuda(), torch.from_numpy(_pack_bits(us)).cuda(), cp, E, out_bits=out)
    dec.synchronize()
    assert np.array_equal(E.cpu().numpy(), ref_E)
    assert np.array_equal(unpack_bits(out.cpu().numpy(), N), ref_uh.astype(np.uint8))
    # what the plain BP kernel decides after the same number of iterations
    uh, _, _ = dec.decode_batch(llr)
    assert np.array_equal(uh, ref_uh)


@pytest.mark.parametrize("K,iters,B,dtype", [(64, 100, 257, "f64"), (20, 7, 5, "f64"), (100, 1, 1, "f64"), (64, 30, 131, "f32"),
                                             (90, 2, 4, "f32")])
def test_bp_register_kernel_n128_vs_oracle_and_lds_kernel(K, iters, B, dtype, oracle):
    """k_bp_w128 (BP at N = 128, one codeword per wavefront, every message in registers; BP_128.c:334-388): the oracle's
    decisions for several rates, iteration counts (1: only one left-going sweep reaches the decision) and batch sizes that
    leave the last workgroup partly filled; the same through the observation input (2*y/std/std formed in the kernel); and
    the generic LDS kernel k_bp (test library) on the same frames."""
    import polardecoding_amd as pa
    from polardecoding_amd import testing as T
    N = 128
    code = oracle.Code(N, K)
    sim = oracle.Sim(4000 + K + iters)
    sig = oracle.sigma_from_db(2.0)
    us, ys = sim.frames(code, sig, B)
    ys = np.stack(ys)
    llr = np.stack([oracle.llr_from_y(y, sig) for y in ys])
    if dtype == "f32":
        llr = llr.astype(np.float32).astype(np.float64)
    ref, _, _ = oracle.decode(code, llr, "BP", bp_iters=iters, dtype=dtype)
    dec = pa.BP(N, K, iterMax=iters, dtype=pa.F64 if dtype == "f64" else pa.F32)
    assert dec.kernel_name.startswith("k_bp_w128")
    uh, _, _ = dec.decode_batch(llr)
    assert np.array_equal(uh, ref)
    if dtype == "f64":
        uy, _, _ = dec.decode_batch_y(ys, sig)
        assert np.array_equal(uy, ref)
    gen = pa.BP(N, K, iterMax=iters, dtype=pa.F64 if dtype == "f64" else pa.F32)
    T.select_kernel(gen, T.KERNEL_GENERIC)
    assert gen.kernel_name.startswith("k_bp<")
    ug, _, _ = gen.decode_batch(llr)
    assert np.array_equal(ug, ref)


@pytest.mark.parametrize("variant", ["FOUR_PER_WAVE", "AUTO", "ONE_PER_WAVE"])
@pytest.mark.parametrize("algo,dtype,B", [("CASCL", "f64", 203), ("SCL", "f64", 64), ("CASCL", "f32", 130), ("CASCL", "f64", 1)])
def test_tuned_list_kernels_vs_oracle(variant, algo, dtype, B, oracle):
    """The three tuned L = 8 kernels for N = 1024 (one, two, four codewords per wavefront) on the same seeded frames:
    decisions, path metric and flags of the oracle; batch sizes that leave the last wavefront partly filled."""
    import polardecoding_amd as pa
    from polardecoding_amd import testing as T
    N, K = 1024, 512
    taps = pa.CRC24C_TAPS if algo == "CASCL" else None
    code = oracle.Code(N, K, taps)
    dt = pa.F64 if dtype == "f64" else pa.F32
    dec = pa.CASCL(N, K, L=8, dtype=dt) if algo == "CASCL" else pa.SCLdecode(N, K, L=8, dtype=dt)
    T.select_kernel(dec, getattr(T, "KERNEL_" + variant))
    sim = oracle.Sim(900 + B)
    llrs = []
    for db in (0.5, 1.5, 2.5):
        sig = oracle.sigma_from_db(db)
        _, ys = sim.frames(code, sig, (B + 2) // 3)
        llrs.append(np.stack([oracle.llr_from_y(y, sig) for y in ys]))
    llr = np.concatenate(llrs)[:B].astype(np.float32).astype(np.float64)
    ref_uh, ref_pm, ref_t = oracle.decode(code, llr, algo, L=8, dtype=dtype)
    uh, pm, fl = dec.decode_batch(llr)
    assert np.array_equal(uh, ref_uh)
    assert np.array_equal(pm.astype(np.float32 if dtype == "f32" else np.float64), ref_pm)
    assert np.array_equal((fl & pa.FLAG_TIE) != 0, ref_t > 0)
