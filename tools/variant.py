#!/usr/bin/env python3
"""Developer tool: build variants of the library for same-box A/B comparisons of one kernel.

    python tools/variant.py build NAME [--tu k_fast2] [-DFOO=1 ...] [--flags "-mllvm ..."]
        compiles the named translation unit(s) with the extra flags and links them with the default objects of the
        other units into build/variants/libpolar_hip_NAME.so (run __graft_entry__.build_library() first).
        With --stamps the host unit is rebuilt with -DPOLAR_STAMPS too (section timers printed at exit).
    python tools/variant.py ab NAME1 NAME2 ... [--config cascl|scl|cfg5|bp|sc] [--reps 5] [--dtype f64]
        on the GPU box: times the configuration's kernel with each library in its own process (same inputs, fixed seed),
        checks the decisions against the first one and against the CPU oracle on 64 frames, prints one line per library.
        "base" names the shipped polardecoding_amd/lib/libpolar_hip.so.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
VDIR = os.path.join(REPO, "build", "variants")


def lib_of(name):
    if name == "base":
        return os.path.join(REPO, "polardecoding_amd", "lib", "libpolar_hip.so")
    return os.path.join(VDIR, f"libpolar_hip_{name}.so")


def build(args, extra):
    import __graft_entry__ as g
    g.build_library(testing=False)
    os.makedirs(VDIR, exist_ok=True)
    odir = os.path.join(g.OBJDIR, "var_" + args.name)
    os.makedirs(odir, exist_ok=True)
    tus = args.tu.split(",")
    if args.stamps and "polar_hip" not in tus:
        tus.append("polar_hip")
        extra = list(extra) + ["-DPOLAR_STAMPS"]
    flags = [f for f in g.HIPCC_FLAGS]
    if args.no_sched:
        flags = [f for f in flags if "sched-strategy" not in f and f != "-mllvm"]
    flags += list(extra) + (args.flags.split() if args.flags else [])
    objs = []
    procs = []
    for tu in g.KERNEL_TUS + ["polar_hip"]:
        if tu in tus:
            obj = os.path.join(odir, tu + ".o")
            cmd = [g._hipcc()] + flags + ["-c", "-o", obj, os.path.join(g.CSRC, tu + ".hip")]
            procs.append((tu, subprocess.Popen(cmd, cwd=g.CSRC, stderr=subprocess.PIPE, text=True)))
            objs.append(obj)
        else:
            objs.append(os.path.join(g.OBJDIR, tu + ".o"))
    for tu, p in procs:
        _, err = p.communicate()
        if p.returncode:
            sys.stderr.write(err[-4000:])
            raise SystemExit(f"compile of {tu} failed")
    out = lib_of(args.name)
    subprocess.check_call([g._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC",
                           "-Wl,--version-script=" + os.path.join(g.CSRC, "polar_hip.map"), "-o", out] + objs)
    if args.meta:
        s = os.path.join(odir, "k.s")
        subprocess.check_call([g._hipcc()] + flags + ["-S", "--cuda-device-only", "-o", s, os.path.join(g.CSRC, tus[0] + ".hip")],
                              cwd=g.CSRC, stderr=subprocess.DEVNULL)
        import re
        txt = open(s).read()
        for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", txt):
            if args.grep in m.group(1):
                print(f"  {m.group(1)}: vgpr {m.group(2)} spill {m.group(3)}")
    print("built", out)


def worker(args):
    """One library, one process: prints a JSON line."""
    import numpy as np
    import torch
    import polardecoding_amd.api as A
    lib = lib_of(args.worker)
    A.lib_path = lambda testing=False: lib
    import polardecoding_amd as pa
    dt = pa.F64 if args.dtype == "f64" else pa.F32
    cfg = {"cascl": (lambda: pa.CASCL(1024, 512, L=8, dtype=dt), 1024, 1 << 17),
           "scl": (lambda: pa.SCLdecode(1024, 512, L=8, dtype=dt), 1024, 1 << 16),
           "cascl128": (lambda: pa.CASCL(128, 64, L=8, crc_taps=pa.CRC6_TAPS, dtype=dt), 128, 1 << 18),
           "cfg5": (lambda: pa.CASCL(4096, 2048, L=32, dtype=dt), 4096, 1 << 15),
           "scl32": (lambda: pa.SCLdecode(1024, 512, L=32, dtype=dt), 1024, 1 << 14),
           "bp": (lambda: pa.BP(1024, 512, iterMax=50, dtype=dt), 1024, 1 << 16),
           "bp128": (lambda: pa.BP(128, 64, iterMax=100, dtype=dt), 128, 1 << 18),
           "sc": (lambda: pa.SCdecode(1024, 512, dtype=dt), 1024, 1 << 18)}[args.config]
    mk, N, B = cfg
    if args.batch:
        B = args.batch
    dec = mk()
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    sigma = 10 ** (-args.snr / 20)
    y = 1.0 + sigma * torch.randn(B, N, dtype=torch.float64, device="cuda", generator=g)
    x = (2 * y / sigma / sigma).to(torch.float64 if args.dtype == "f64" else torch.float32).contiguous()
    del y
    out = torch.empty(B, N // 32, dtype=torch.int32, device="cuda")
    pm = torch.empty(B, dtype=torch.float64, device="cuda")
    fl = torch.empty(B, dtype=torch.int32, device="cuda")
    dec.decode_device(x, out_bits=out, pm=pm, flags=fl)
    dec.synchronize()
    h = hashlib.sha256(out.cpu().numpy().tobytes() + pm.cpu().numpy().tobytes() + fl.cpu().numpy().tobytes()).hexdigest()[:16]
    times = [dec.time_decode_device(x, out, 1) for _ in range(args.reps)]
    times.sort()
    oracle_ok = None
    if args.oracle and args.dtype == "f64":
        from oracle import oracle_py as O
        nf = 48 if N <= 1024 else 6
        llr = x[:nf].cpu().numpy().astype(np.float64)
        algo = {pa.ALGO_SC: "SC", pa.ALGO_BP: "BP", pa.ALGO_SCL: "SCL", pa.ALGO_CASCL: "CASCL"}[dec.algo]
        taps = None
        if dec.algo == pa.ALGO_CASCL:
            taps = O.CRC6_TAPS if N == 128 else O.CRC24C_TAPS
        if N > 1024:   # no 5G table above 1024: the library's own order (beta expansion), frozen positions first
            io = [int(v) for v in dec.info_order]
            rest = [j for j in range(N) if j not in set(io)]
            code = O.Code(N, dec.K, taps, Q=rest + io)
        else:
            code = O.Code(N, dec.K, taps)
        ref = O.decode(code, llr, algo, L=dec.L, bp_iters=(50 if N == 1024 else 100))[0]
        w = out[:nf].cpu().numpy().view(np.uint32)
        got = ((w[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(nf, N)
        oracle_ok = bool(np.array_equal(got, ref))
    print(json.dumps({"lib": args.worker, "config": args.config, "dtype": args.dtype, "kernel": dec.kernel_name, "frames": B,
                      "ms_min": times[0], "ms_med": times[len(times) // 2], "Mfps": B / times[len(times) // 2] / 1e3,
                      "hash": h, "oracle_ok": oracle_ok}), flush=True)


def ab(args):
    first = None
    rows = []
    for rnd in range(args.rounds):
        for name in args.names:
            if not os.path.exists(lib_of(name)):
                print(f"{name}: {lib_of(name)} missing")
                continue
            cmd = [sys.executable, os.path.abspath(__file__), "ab", "--worker", name, "--config", args.config, "--reps", str(args.reps),
                   "--dtype", args.dtype, "--snr", str(args.snr), "--batch", str(args.batch)] + (["--oracle"] if args.oracle and rnd == 0 else [])
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode or not line:
                print(f"{name}: FAILED rc={r.returncode} {r.stderr[-600:]}")
                continue
            d = json.loads(line[-1])
            if first is None:
                first = d["hash"]
            d["same_as_first"] = d["hash"] == first
            rows.append(d)
            st = [l for l in r.stderr.splitlines() if l.startswith("[stamps]")]
            print(f"{name:24s} {d['config']:8s} {d['dtype']} med {d['ms_med']:8.3f} ms  min {d['ms_min']:8.3f} ms  {d['Mfps']:8.3f} M/s  "
                  f"same={d['same_as_first']} oracle={d['oracle_ok']}  {d['kernel']}", flush=True)
            for l in st:
                print("    " + l)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["build", "ab"])
    ap.add_argument("names", nargs="*")
    ap.add_argument("--tu", default="k_fast2")
    ap.add_argument("--flags", default="")
    ap.add_argument("--stamps", action="store_true")
    ap.add_argument("--no-sched", action="store_true", help="drop the -amdgpu-sched-strategy flag")
    ap.add_argument("--meta", action="store_true", help="print vgpr / spill counts of the unit's kernels")
    ap.add_argument("--grep", default="", help="with --meta: only kernels whose mangled name contains this")
    ap.add_argument("--config", default="cascl")
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=1)
    ap.add_argument("--snr", type=float, default=2.0)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--worker", default="")
    args, extra = ap.parse_known_args()
    if args.cmd == "build":
        args.name = args.names[0]
        build(args, extra)
    elif args.worker:
        worker(args)
    else:
        ab(args)


if __name__ == "__main__":
    main()
