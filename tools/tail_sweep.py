"""Single-launch rate of the headline kernel against the batch size: how much of a launch is head + tail?
B = k x 6144 fills every resident wavefront (768 workgroups x 4 wavefronts x 2 codewords) exactly k times."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import torch
import polardecoding_amd as pa
sigma = 10 ** (-2.0 / 20)
rng = np.random.default_rng(3)
dec = pa.CASCL(1024, 512, L=8)
Bmax = 6144 * 44
x = torch.from_numpy(2 * (1.0 + sigma * rng.standard_normal((Bmax, 1024))) / sigma / sigma).cuda()
ob = torch.empty(Bmax, 32, dtype=torch.int32, device="cuda")
dec.decode_device(x[:6144], out_bits=ob[:6144]); dec.synchronize()
for B in (6144, 2 * 6144, 4 * 6144, 10 * 6144, 20 * 6144, 21 * 6144, 131072, 22 * 6144, 43 * 6144, 262144, 44 * 6144):
    ms = min(dec.time_decode_device(x[:B], ob[:B], 3) for _ in range(3))
    print(f"B={B:7d} ({B/6144:6.2f} rounds) {ms:8.3f} ms  {B/ms/1e3:7.3f} M frames/s  {ms/ (B/6144):7.4f} ms/round", flush=True)
