import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import torch
import polardecoding_amd as pa
from polardecoding_amd import testing as T
sigma = 10 ** (-2.0 / 20)
rng = np.random.default_rng(3)
for var, name in ((0, "fast2 (default)"), (4, "fast (one per wave)")):
    dec = pa.CASCL(1024, 512, L=8)
    if var: T.select_kernel(dec, var)
    y = 1.0 + sigma * rng.standard_normal(1024)
    for _ in range(50): dec(y, sigma)
    lat = []
    for _ in range(1000):
        t0 = time.perf_counter(); dec(y, sigma); lat.append(time.perf_counter() - t0)
    lat.sort()
    out = [f"{name:22s} {dec.kernel_name:32s} polar_decode median {lat[500]*1e6:7.1f} us"]
    for B in (2, 16, 128, 1024, 4096):
        x = torch.from_numpy(2 * (1.0 + sigma * rng.standard_normal((B, 1024))) / sigma / sigma).cuda()
        ob = torch.empty(B, 32, dtype=torch.int32, device="cuda")
        dec.decode_device(x, out_bits=ob); dec.synchronize()
        ms = dec.time_decode_device(x, ob, 20)
        out.append(f"B={B}: {ms*1e3:7.1f} us")
    print("  ".join(out), flush=True)
