"""Developer tool: k_scl_fast2 (two codewords per wavefront) against k_scl_fast4 (four) over code rates, N = 1024 L = 8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import polardecoding_amd as pa
from polardecoding_amd import testing as T
B = 1 << 17
sigma = 10 ** (-2.0 / 20)
y = 1.0 + sigma * torch.randn(B, 1024, dtype=torch.float64, device="cuda")
llr = (2 * y / sigma / sigma).contiguous()
out = torch.empty(B, 32, dtype=torch.int32, device="cuda")
for K in (128, 256, 512, 640, 768, 896, 1000):
    row = []
    for var in (T.KERNEL_AUTO, T.KERNEL_FOUR_PER_WAVE):
        dec = pa.SCLdecode(1024, K, L=8)
        T.select_kernel(dec, var)
        dec.decode_device(llr, out_bits=out); dec.synchronize()
        row.append(dec.time_decode_device(llr, out, 3))
    print(f"K={K:5d}  fast2 {row[0]:7.3f} ms {B/row[0]/1e3:6.2f} M/s   fast4 {row[1]:7.3f} ms {B/row[1]/1e3:6.2f} M/s   fast4/fast2 speed {row[0]/row[1]:.3f}", flush=True)
