"""Per-octet-type cost experiment (developer tool): decode random LLRs with artificial frozen sets."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import polardecoding_amd as pa
N = 1024
B = 32768
llr = torch.randn(B, N, dtype=torch.float64, device="cuda") * 2.5 + 3.0
out = torch.empty(B, N // 32, dtype=torch.int32, device="cuda")
def run(name, info, dtype=pa.F64):
    dec = pa.Decoder(N, len(info), pa.ALGO_SCL, L=8, info_order=np.array(info, dtype=np.int32), dtype=dtype)
    x = llr if dtype == pa.F64 else llr.float()
    dec.decode_device(x, out_bits=out); dec.synchronize()
    ms = dec.time_decode_device(x, out, 3)
    print(f"{name:28s} K={len(info):5d}  {ms:8.3f} ms  {ms*1e3/B*2048:9.1f} us per frame-wave -> {B/ms/1e3:7.2f} Mframes/s")
for dt, nm in ((pa.F64, "f64"), (pa.F32, "f32")):
    print(nm)
    run("last 8 info (127 frozen oct)", list(range(N - 8, N)), dt)
    run("last octet + 0x7F everywhere", [8 * o + 7 for o in range(127)] + list(range(N - 8, N)), dt)
    run("all info but first 8", list(range(8, N)), dt)
    run("0x01 everywhere (SPC-like)", [j for j in range(N) if j % 8 != 0], dt)
    run("0x17 everywhere", [j for j in range(N) if (j % 8) in (3, 5, 6, 7)], dt)
    run("5G K=512", pa.q_sequence(N)[N - 512:], dt)
