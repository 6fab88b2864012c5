#!/usr/bin/env python3
"""FER-vs-Eb/N0 of the GPU decoders next to the reference's published BLER (tests/golden/published_runs.json =
the reference's myResult_*.zip logs).  Throughput mode: frames generated on the device (polar_generate_device,
Philox), decoded, compared on the device (polar_fer_batch via `polar_sim --fast`); the stop rule is "at least
`ble` block errors, whole batches".  The GPU points carry many more errors than the published ones, so the
published values should lie within THEIR binomial error of the GPU curve.

    python tools/fer_curves.py > profiles/rNN_fer_curves.txt
"""
import json, math, os, re, subprocess, sys, time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = os.path.join(REPO, "polardecoding_amd", "lib", "polar_sim")
PUB = json.load(open(os.path.join(REPO, "tests", "golden", "published_runs.json")))

CURVES = [
    # title, published key, L, args, snr range, ble, batch
    ("SC N=1024 K=512", "myResult_1024/SC1024out.dat", 1, ["--algo", "sc", "--N", "1024", "--K", "512"], "1.0:4.0:0.5", 3000, 1 << 17),
    ("SCL N=1024 K=512 L=8", "myResult_1024/SCL1024out.dat", 8, ["--algo", "scl", "--N", "1024", "--K", "512", "--L", "8"], "1.0:3.0:0.5", 3000, 1 << 16),
    ("CA-SCL N=1024 K=512 CRC-24C L=8", "myResult_1024/CASCL_L8.dat", 8, ["--algo", "cascl", "--N", "1024", "--K", "512", "--L", "8", "--crc", "24c"], "1.0:2.5:0.5", 3000, 1 << 17),
    ("CA-SCL N=128 K=64 CRC-6 L=8", "myResult_128/CASCL_128_L8.txt", 8, ["--algo", "cascl", "--N", "128", "--K", "64", "--L", "8", "--crc", "6"], "1.0:3.5:0.5", 5000, 1 << 17),
    ("SCL N=128 K=64 L=8", "myResult_128/SCL128out_errblock50.dat", 8, ["--algo", "scl", "--N", "128", "--K", "64", "--L", "8"], "1.0:3.5:0.5", 5000, 1 << 17),
    ("CA-SCL N=1024 K=512 CRC-24C L=32 (k_scl_big)", "myResult_1024/CASCL_L32.dat", 32, ["--algo", "cascl", "--N", "1024", "--K", "512", "--L", "32", "--crc", "24c"], "1.0,1.5,2.0,2.2", 2000, 1 << 15),
    ("CA-SCL N=128 K=64 CRC-6 L=32", "myResult_128/CASCL_128_L32.txt", 32, ["--algo", "cascl", "--N", "128", "--K", "64", "--L", "32", "--crc", "6"], "1.0:3.5:0.5", 3000, 1 << 16),
    # BP: the seed-labelled log myResult_1024.zip:BP1024out_NewSEED.dat (SEED 771, 200 errors per point, iterMax 100)
    ("BP N=1024 K=512, 100 iterations (k_bp_r4)", {1.0: (200, 445), 1.5: (200, 1294), 2.0: (200, 6076), 2.5: (200, 35242), 3.0: (200, 162847), 3.5: (200, 920196)}, 1,
     ["--algo", "bp", "--N", "1024", "--K", "512", "--bp-iters", "100"], "1.0:3.5:0.5", 2000, 1 << 16),
    # BP N=128: myResult_128.zip:BP128_BER.txt (SEED 834, 200 errors per point, iterMax 100); runs = 200 / published BLER
    ("BP N=128 K=64, 100 iterations (k_bp_w128)", {1.0: (200, 449), 1.5: (200, 833), 2.0: (200, 1573), 2.5: (200, 3887), 3.0: (200, 12658), 3.5: (200, 37594), 4.0: (200, 98039)}, 1,
     ["--algo", "bp", "--N", "128", "--K", "64", "--bp-iters", "100"], "1.0:4.0:0.5", 3000, 1 << 17),
]


def published(key, L):
    pts = {}
    if isinstance(key, dict):
        return dict(key)
    for b in PUB[key]:
        if b["L"] != L:
            continue
        for snr, eb, run in b["rows"]:
            e, r = pts.get(snr, (0, 0))
            pts[snr] = (e + eb, r + run)
    return pts


for title, key, L, args, snr, ble, batch in CURVES:
    t0 = time.time()
    out = subprocess.run([SIM] + args + ["--fast", "--snr-list" if "," in snr else "--snr", snr, "--ble", str(ble), "--batch", str(batch), "--seed", "20261004"],
                         capture_output=True, text=True, timeout=1500)
    dt = time.time() - t0
    if out.returncode:
        print(title, "FAILED", out.stderr[-500:])
        continue
    pub = published(key, L)
    print(f"## {title}   (polar_sim --fast, {dt:.1f} s wall for the whole sweep)")
    print("Eb/N0  frames        block-errors  FER(GPU)     +-95%      published(errors/frames)  published BLER  |dev|/sigma_pub")
    tot = 0
    for m in re.finditer(r"bSNR = ([\d.]+)\terror block = (\d+)\trun = (\d+)", out.stdout):
        s, e, r = float(m.group(1)), int(m.group(2)), int(m.group(3))
        tot += r
        fer = e / r
        ci = 1.96 * math.sqrt(fer * (1 - fer) / r)
        line = f"{s:4.1f}  {r:12d}  {e:12d}  {fer:.4e}  {ci:.1e}"
        if s in pub:
            pe, pr = pub[s]
            pf = pe / pr
            sig = math.sqrt(fer * (1 - fer) / pr)      # binomial error of the published estimate around the GPU value
            line += f"  {pe:6d}/{pr:<10d}          {pf:.4e}     {abs(pf - fer) / sig:5.2f}"
        print(line)
    print(f"# {tot} frames generated + decoded + compared in {dt:.1f} s = {tot / dt / 1e6:.2f} M frames/s end to end (process start-up included)\n")
    sys.stdout.flush()
