#!/usr/bin/env python3
"""Differential sweep GPU (through the C ABI) vs the CPU oracle over many shapes: every kernel family, list sizes,
code lengths, rates, CRCs, both arithmetic types, ragged batch sizes.  Developer tool (tests/ holds the fixed cases)."""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polardecoding_amd as pa
from oracle import oracle_py as O

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2026)   # optional argument: another seed (other batch sizes and frames)
bad = 0
t0 = time.time()


def q_of(dec, N, K, taps):
    io = dec.info_order.tolist()
    s = set(io)
    return [j for j in range(N) if j not in s] + io


def check(tag, dec, code, algo, L, B, db, dtype, iters=20):
    global bad
    sim = O.Sim(int(rng.integers(1, 1 << 30)))
    sig = O.sigma_from_db(db)
    us, ys = sim.frames(code, sig, B)
    llr = np.stack([O.llr_from_y(y, sig) for y in ys]).astype(np.float32).astype(np.float64)
    ref, rpm, _ = O.decode(code, llr, algo, L=L, bp_iters=iters, dtype=dtype)
    uh, pm, fl = dec.decode_batch(llr)
    ok = np.array_equal(uh, ref) and (dtype == "f32" or algo in ("SC", "BP") or np.array_equal(pm, rpm))
    nerr = int((uh != us).any(axis=1).sum())
    print(f"{'ok ' if ok else 'BAD'} {tag:48s} {dec.kernel_name[:34]:34s} B={B:4d} frames-in-error={nerr}", flush=True)
    bad += (not ok)


for dtype in ("f64", "f32"):
    dt = pa.F64 if dtype == "f64" else pa.F32
    # list decoders
    for N, L in itertools.product((512, 1024, 2048, 4096), (2, 4, 8, 16, 32)):
        for K, taps in ((N // 2, pa.CRC24C_TAPS), (N // 4, None), (3 * N // 4, pa.CRC6_TAPS)):
            if N == 4096 and L >= 16 and K != N // 2:
                continue
            dec = pa.CASCL(N, K, L=L, crc_taps=taps, dtype=dt) if taps else pa.SCLdecode(N, K, L=L, dtype=dt)
            code = O.Code(N, K, taps, Q=q_of(dec, N, K, taps))
            B = int(rng.integers(3, 9)) if N * L >= 32768 else int(rng.integers(5, 20))
            check(f"{'CASCL' if taps else 'SCL'} N={N} K={K} L={L} r={max(taps) if taps else 0} {dtype}", dec, code,
                  "CASCL" if taps else "SCL", L, B, 1.5 if K * 2 <= N else 3.5, dtype)
    for N, L in itertools.product((32, 64, 128, 256), (1, 2, 8, 32)):
        K = N // 2
        dec = pa.SCLdecode(N, K, L=L, dtype=dt)
        code = O.Code(N, K, None, Q=q_of(dec, N, K, None))
        check(f"SCL N={N} K={K} L={L} {dtype}", dec, code, "SCL", L, int(rng.integers(5, 40)), 2.0, dtype)
    # SC: lanes kernel (B >= 64) and the generic one (B < 64)
    for N in (32, 64, 128, 256, 512, 1024, 2048):
        for K in (max(1, N // 8), N // 2, N - N // 8):
            dec = pa.SCdecode(N, K, dtype=dt)
            code = O.Code(N, K, None, Q=q_of(dec, N, K, None))
            for B in (int(rng.integers(1, 63)), int(rng.integers(64, 200))):
                check(f"SC N={N} K={K} {dtype}", dec, code, "SC", 1, B, 2.0 if K * 2 <= N else 5.0, dtype)
    # the four tuned L = 8 kernels for N = 1024 side by side (one, two, four codewords per wavefront; big-list kernel)
    from polardecoding_amd import testing as T
    for variant in ("AUTO", "ONE_PER_WAVE", "FOUR_PER_WAVE", "BIG"):
        for K, taps in ((512, pa.CRC24C_TAPS), (256, None), (768, pa.CRC6_TAPS), (1000, None), (24, None)):
            dec = pa.CASCL(1024, K, L=8, crc_taps=taps, dtype=dt) if taps else pa.SCLdecode(1024, K, L=8, dtype=dt)
            T.select_kernel(dec, getattr(T, "KERNEL_" + variant))
            code = O.Code(1024, K, taps, Q=q_of(dec, 1024, K, taps))
            check(f"{variant} N=1024 K={K} r={max(taps) if taps else 0} {dtype}", dec, code, "CASCL" if taps else "SCL", 8,
                  int(rng.integers(1, 40)), 1.5 if K <= 512 else 4.5, dtype)
    # BP (N = 1024: the register-blocked kernel, several rates and iteration counts)
    for N, K, it in ((32, 16, 7), (128, 64, 20), (512, 256, 11), (1024, 512, 6), (1024, 200, 13), (1024, 900, 50), (2048, 1024, 4),
                     (4096, 2048, 3)):   # above 1024: rows in global scratch
        dec = pa.BP(N, K, iterMax=it, dtype=dt)
        code = O.Code(N, K, None, Q=q_of(dec, N, K, None))
        check(f"BP N={N} K={K} it={it} {dtype}", dec, code, "BP", 1, int(rng.integers(3, 12)), 2.0 if K * 2 <= N else 5.0, dtype, iters=it)
print(f"{bad} mismatching configurations, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
