#!/usr/bin/env python3
"""Throughput of every BASELINE.json config on one GPU (developer tool; bench.py is the contract benchmark).
Inputs: BPSK-AWGN LLRs of the all-zero codeword at 2 dB (valid for every linear code), resident in HBM."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import polardecoding_amd as pa

ap = argparse.ArgumentParser()
ap.add_argument("--only", default="")
ap.add_argument("--dtype", default="f64")
ap.add_argument("--variant", type=int, default=0, help="polar_hip_testing.h kernel variant (0 = the library's choice)")
args = ap.parse_args()
dt = pa.F64 if args.dtype == "f64" else pa.F32
tdt = torch.float64 if args.dtype == "f64" else torch.float32
sigma = 10 ** (-2.0 / 20)

def llrs(B, N):
    y = 1.0 + sigma * torch.randn(B, N, dtype=torch.float64, device="cuda")
    return (2 * y / sigma / sigma).to(tdt).contiguous()

CONFIGS = [
    ("SC_1024", lambda: pa.SCdecode(1024, 512, dtype=dt), 1024, 1 << 18),
    ("BP_1024_50it", lambda: pa.BP(1024, 512, iterMax=50, dtype=dt), 1024, 1 << 16),
    ("BP_128_100it", lambda: pa.BP(128, 64, iterMax=100, dtype=dt), 128, 1 << 18),
    ("SCL_1024_L8", lambda: pa.SCLdecode(1024, 512, L=8, dtype=dt), 1024, 1 << 16),
    ("CASCL_1024_L8", lambda: pa.CASCL(1024, 512, L=8, dtype=dt), 1024, 1 << 17),
    ("CASCL_128_L8", lambda: pa.CASCL(128, 64, L=8, crc_taps=pa.CRC6_TAPS, dtype=dt), 128, 1 << 18),
    ("SCL_1024_L32", lambda: pa.SCLdecode(1024, 512, L=32, dtype=dt), 1024, 1 << 14),
    ("CASCL_4096_L32", lambda: pa.CASCL(4096, 2048, L=32, dtype=dt), 4096, 1 << 15),
]
for name, mk, N, B in CONFIGS:
    if args.only and args.only not in name:
        continue
    dec = mk()
    if args.variant:
        from polardecoding_amd import testing as T
        T.select_kernel(dec, args.variant)
    x = llrs(B, N)
    out = torch.empty(B, N // 32, dtype=torch.int32, device="cuda")
    dec.decode_device(x, out_bits=out); dec.synchronize()
    ms = dec.time_decode_device(x, out, 2)
    nerr = int((out != 0).any(dim=1).sum().item())
    print(json.dumps({"config": name, "dtype": args.dtype, "kernel": dec.kernel_name, "frames": B, "ms": ms,
                      "frames_per_s": B / ms * 1e3, "frames_in_error": nerr}), flush=True)
