#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point polar_decode_batch (host LLRs in, int u_hat[B][N] out):
what a caller that keeps the reference's per-frame arrays gets.  Not bench.py's `value` (that is HBM-resident)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polardecoding_amd as pa

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1 << 17)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
rng = np.random.default_rng(1)
sigma = 10 ** (-2.0 / 20)
for name, dec, N in (("CASCL_1024_L8", pa.CASCL(1024, 512, L=8), 1024), ("SC_1024", pa.SCdecode(1024, 512), 1024)):
    y = 1.0 + sigma * rng.standard_normal((args.frames, N))
    llr = np.ascontiguousarray(2 * y / sigma / sigma)
    dec.decode_batch(llr[:1024])
    t0 = time.perf_counter()
    for _ in range(args.reps):
        uh, pm, fl = dec.decode_batch(llr)
    dt = (time.perf_counter() - t0) / args.reps
    out = np.zeros((args.frames, N), dtype=np.int32)   # the caller's own, already touched, output array (the reference's u_hat)
    dec.decode_batch(llr, out=out)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        dec.decode_batch(llr, out=out)
    dt2 = (time.perf_counter() - t0) / args.reps
    print(json.dumps({"config": name, "frames": args.frames, "s_per_call_fresh_output_array": dt,
                      "frames_per_s_fresh_output_array": args.frames / dt,
                      "s_per_call": dt2, "frames_per_s_host_buffers": args.frames / dt2,
                      "frames_in_error": int(uh.any(axis=1).sum()), "same": bool((out == uh).all())}), flush=True)
