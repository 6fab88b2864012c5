#!/usr/bin/env python3
"""Developer tool: k_scl_big storage splits side by side (polar_testing_big_split): every split must give the outputs
and path metrics of the first one bit for bit; prints the kernel-timed rate of each.
    python tools/big_split_sweep.py --dtype f64 --splits 35,351,371"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import polardecoding_amd as pa
from polardecoding_amd import testing as T

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f64")
ap.add_argument("--splits", default="35,351,371")
ap.add_argument("--only", default="")
ap.add_argument("--snr", type=float, default=2.0)
args = ap.parse_args()
dt = pa.F64 if args.dtype == "f64" else pa.F32
tdt = torch.float64 if args.dtype == "f64" else torch.float32
sigma = 10 ** (-args.snr / 20)
torch.manual_seed(7)
CONFIGS = [
    ("SCL_1024_L32", lambda: pa.SCLdecode(1024, 512, L=32, dtype=dt), 1024, 1 << 14),
    ("CASCL_1024_L32", lambda: pa.CASCL(1024, 512, L=32, dtype=dt), 1024, 1 << 14),
    ("CASCL_4096_L32", lambda: pa.CASCL(4096, 2048, L=32, dtype=dt), 4096, 1 << 15),
]
for name, mk, N, B in CONFIGS:
    if args.only and args.only not in name:
        continue
    y = 1.0 + sigma * torch.randn(B, N, dtype=torch.float64, device="cuda")
    x = (2 * y / sigma / sigma).to(tdt).contiguous()
    ref = None
    for sp in [int(v) for v in args.splits.split(",")]:
        dec = mk()
        T.big_split(dec, sp)
        out = torch.empty(B, N // 32, dtype=torch.int32, device="cuda")
        pm = torch.empty(B, dtype=torch.float64, device="cuda")
        fl = torch.empty(B, dtype=torch.int32, device="cuda")
        dec.decode_device(x, out_bits=out, pm=pm, flags=fl)
        dec.synchronize()
        ms = dec.time_decode_device(x, out, 2)
        same = None
        if ref is None:
            ref = (out.clone(), pm.clone(), fl.clone())
        else:
            same = bool((out == ref[0]).all().item() and (pm == ref[1]).all().item() and (fl == ref[2]).all().item())
        print(json.dumps({"config": name, "dtype": args.dtype, "split": sp, "frames": B, "ms": round(ms, 3),
                          "frames_per_s": round(B / ms * 1e3), "same_as_first": same,
                          "frames_in_error": int((out != 0).any(dim=1).sum().item())}), flush=True)
        del dec
