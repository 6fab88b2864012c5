"""Developer tool: decode the same batch with two builds of the library in ONE process and list the frames whose results
differ (decisions / path metric / flags), with the oracle's verdict on them.
    python tools/diff_libs.py build/variants/libpolar_hip_X.so [--config cfg5] [--frames 32768]"""
import argparse, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import polardecoding_amd as pa
import polardecoding_amd.api as A
from polardecoding_amd.synth import make_batch

ap = argparse.ArgumentParser()
ap.add_argument("lib")
ap.add_argument("--frames", type=int, default=32768)
ap.add_argument("--snr", type=float, default=1.5)
args = ap.parse_args()
La = A.load_library()
A._libs.clear()
other = os.path.abspath(args.lib)
A.lib_path = lambda testing=False: other
Lb = A.load_library()
N, K, L = 4096, 2048, 32
taps = pa.CRC24C_TAPS
decs = [pa.CASCL(N, K, L=L, crc_taps=taps, _library=La), pa.CASCL(N, K, L=L, crc_taps=taps, _library=Lb)]
info = torch.tensor(decs[0].info_order.astype(np.int64), device="cuda")
gen = torch.Generator(device="cuda"); gen.manual_seed(77)
B = args.frames
llr, _ = make_batch(B, N, K, taps, args.snr, info, "cuda", gen)
res = []
for d in decs:
    d.use_torch_stream()
    pm = torch.zeros(B, dtype=torch.float64, device="cuda"); fl = torch.zeros(B, dtype=torch.int32, device="cuda")
    bits = d.decode_device(llr, pm=pm, flags=fl); torch.cuda.synchronize()
    res.append((bits.clone(), pm, fl))
    print(d.kernel_name)
db = (res[0][0] != res[1][0]).any(dim=1); dp = res[0][1].view(torch.int64) != res[1][1].view(torch.int64); df = res[0][2] != res[1][2]
bad = torch.nonzero(db | dp | df).flatten().tolist()
print(f"{len(bad)} of {B} frames differ: bits {int(db.sum())}, pm {int(dp.sum())}, flags {int(df.sum())}")
bb = torch.nonzero(db).flatten().tolist()
print("first frames with different decisions:", bb[:24])
print("  of them below 3072 (first job of a wavefront):", sum(1 for f in bb if f < 3072), " below 6144:", sum(1 for f in bb if f < 6144))
fx = (res[0][2] ^ res[1][2])
print("flag bits that differ (bit: frames):", {b: int(((fx >> b) & 1).sum()) for b in range(4)})
if bad:
    from oracle import oracle_py as O
    io = decs[0].info_order
    rest = [j for j in range(N) if j not in set(io.tolist())]
    code = O.Code(N, K, taps, Q=rest + io.tolist())
    for f in bad[:6]:
        uh, pmo, _ = O.decode(code, llr[f].cpu().numpy(), "CASCL", L=L)
        for name, r in zip(("first", "second"), res):
            w = r[0][f].cpu().numpy().view(np.uint32)
            got = ((w[:, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(N)
            print(f"frame {f} {name}: bits==oracle {bool(np.array_equal(got, np.reshape(uh, (-1, N))[0]))} pm {float(r[1][f])!r} (oracle {float(np.ravel(pmo)[0])!r}) flags {int(r[2][f])}")
