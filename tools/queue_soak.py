"""Developer tool: many queued launches in a row on one context, every result compared with the first one (a lost or doubly
decoded job, or a counter that is not back at zero, would show as a different or missing frame)."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import torch
import polardecoding_amd as pa
from polardecoding_amd.synth import make_batch
CRC = pa.CRC24C_TAPS
cases = [("CASCL_1024_L8", lambda: pa.CASCL(1024, 512, L=8, crc_taps=CRC), 1024, 512, CRC, [131072, 6144 * 3 + 1, 70001, 6145], 120),
         ("BP_1024_10it", lambda: pa.BP(1024, 512, iterMax=10), 1024, 512, None, [65536, 769, 5000], 60),
         ("CASCL_128_L8", lambda: pa.CASCL(128, 64, L=8, crc_taps=pa.CRC6_TAPS), 128, 64, pa.CRC6_TAPS, [262144, 99999], 100),
         ("CASCL_4096_L32", lambda: pa.CASCL(4096, 2048, L=32, crc_taps=CRC), 4096, 2048, CRC, [8192, 3073], 6)]
for name, mk, N, K, taps, sizes, reps in cases:
    dec = mk(); dec.use_torch_stream()
    info = torch.tensor(dec.info_order.astype(np.int64), device="cuda")
    gen = torch.Generator(device="cuda"); gen.manual_seed(123)
    llr, _ = make_batch(max(sizes), N, K, taps, 1.5, info, "cuda", gen)
    ref = {}
    bad = 0
    n = 0
    for r in range(reps):
        for B in sizes:
            out = torch.full((B, N // 32), -1, dtype=torch.int32, device="cuda")
            dec.decode_device(llr[:B], out_bits=out)
            if B not in ref:
                ref[B] = out.clone()
            elif not torch.equal(out, ref[B]):
                bad += 1
            n += 1
    torch.cuda.synchronize()
    print(f"{name:16s} {n:5d} launches, sizes {sizes}: {bad} differ from the first of their size; undecoded frames in the references: "
          f"{sum(int((v == -1).all(dim=1).sum()) for v in ref.values())}", flush=True)
