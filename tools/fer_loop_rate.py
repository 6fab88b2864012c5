"""Developer tool: rate of the whole Monte-Carlo loop body on the device (polar_fer_batch: generate -> decode -> count)
against the decode alone, and of the generator alone (polar_generate_device)."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
import polardecoding_amd as pa
for name, mk, B in (("CASCL_1024_L8", lambda: pa.CASCL(1024, 512, L=8), 1 << 18), ("BP_1024_50it", lambda: pa.BP(1024, 512, iterMax=50), 1 << 16),
                    ("SC_1024", lambda: pa.SCdecode(1024, 512), 1 << 19), ("CASCL_128_L8", lambda: pa.CASCL(128, 64, L=8, crc_taps=pa.CRC6_TAPS), 1 << 20)):
    dec = mk()
    N = dec.N
    dec.fer_batch(1, 0, 2.0, B)
    t0 = time.perf_counter()
    reps = 3
    for r in range(reps):
        dec.fer_batch(1, (r + 1) * B, 2.0, B)
    t_loop = (time.perf_counter() - t0) / reps
    out = torch.empty(B, N, dtype=torch.float64, device="cuda")
    ub = torch.empty(B, N // 32, dtype=torch.int32, device="cuda")
    dec.generate_device(1, 0, 2.0, out, ub); dec.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        dec.generate_device(1, r * B, 2.0, out, ub)
    dec.synchronize()
    t_gen = (time.perf_counter() - t0) / reps
    bits = torch.empty(B, N // 32, dtype=torch.int32, device="cuda")
    ms = dec.time_decode_device(out, bits, 3)
    print(f"{name:14s} B={B:8d}  loop {B/t_loop/1e6:8.2f} M frames/s ({t_loop*1e3:7.2f} ms)   generator alone {B/t_gen/1e6:8.2f} M ({t_gen*1e3:6.2f} ms)   decode alone {B/ms/1e3:8.2f} M ({ms:6.2f} ms)", flush=True)
