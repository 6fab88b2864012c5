#!/usr/bin/env python3
"""bench.py -- headline benchmark: decoded frames/s of CRC-aided SCL, N=1024 K=512 CRC-24C L=8.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cascl_1024_l8|cascl_4096_l32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

--config cascl_4096_l32 is BASELINE.json's config 5 (N=4096 K=2048 CA-SCL L=32, 2^15 frames per GPU per step = 2^18
over 8 GPUs); the default line (config 4) is unchanged by it.

Started without a launcher and with --gpus N > 1, the process that parsed the arguments touches no GPU: it starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child, passes rank 0's JSON line through and
exits with the child's code.  Started by a launcher, --gpus must equal WORLD_SIZE.

A "step" is one pass of the decode hot path (one kernel launch through the C ABI,
polar_decode_device) over one batch of synthetic BPSK-AWGN frames per GPU, inputs (channel LLRs)
already resident in HBM.  Steps alternate between two contexts / HIP streams per GPU, so the last,
partly filled pass of one launch overlaps the first pass of the next (--one-stream turns that off).  Frames are independent, so the batch shards across ranks with no
data-path collective (weak scaling: 2^17 frames per GPU per step = BASELINE config 4's 2^20 over
8 GPUs); RCCL is used only for the final block/bit error counters and the max-over-ranks time.

Rank 0 prints ONE JSON line with the contract fields plus `roofline` (dominant kernel, HIP-event
timed on its stream) and `cpu_baseline` (the real reference C, oracle/_ref, one thread, bounded
sample).  The CPU oracle / reference is imported ONLY for that baseline leg.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# torch and the HIP library are imported by the worker only (run()): the launching parent must never initialise a GPU
CRC = (0, 1, 2, 4, 8, 12, 13, 15, 17, 20, 21, 23, 24)   # CRC-24C, CASCL_1024_L8.c:2-4 (= polardecoding_amd.CRC24C_TAPS)
R = max(CRC)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
VALU_PEAK_LANE_OPS = 3.9e13   # 256 CUs x 64 lanes x 2.4 GHz, non-packed VALU
TRAFFIC_PROFILE = "r03_traffic.json"   # latest committed rocprofv3 --pmc summary of the dominant kernels
# BASELINE.md section 2 (survey-measured, one core, decode call only; the reference publishes no throughput): frames/s
BASELINE_MD_CPU = {"cascl_1024_l8": 138.5}

# The workloads bench.py can run as N-rank lines.  cascl_1024_l8 = BASELINE.json configs[3] (the one `metric` is quoted on);
# cascl_4096_l32 = configs[4] ("LLRs spill HBM", 2^18 frames over 8 GPUs).  lane_ops: SURVEY 8d's algorithmic VALU count per
# frame -- L (N/2) n CHK at 25..40 lane-ops each + as many g (1) + N L PHI (~8) + (K+r) sixteen-way (2L-way) rankings + CRC.
CONFIGS = {
    "cascl_1024_l8": dict(N=1024, K=512, L=8, batch=1 << 17, metric="decoded frames/sec, N=1024 K=512 CA-SCL L=8",
                          label="CASCL_1024_L8: N=1024 K=512 CRC-24C L=8", over8="2^20 over 8 GPUs",
                          lane_ops=(1.2e6, 1.8e6), traffic_key=None),
    "cascl_4096_l32": dict(N=4096, K=2048, L=32, batch=1 << 15, metric="decoded frames/sec, N=4096 K=2048 CA-SCL L=32",
                           label="CASCL_4096_L32: N=4096 K=2048 CRC-24C L=32 (5G-eMBB-style, LLR levels spill HBM)",
                           over8="2^18 over 8 GPUs", lane_ops=(2.5e7, 3.9e7), traffic_key="cfg5"),
}
N, K = 1024, 512   # the headline shape (cpu_baseline, fer_sweep)


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def fer_vs_snr(dec, frames):
    """The second half of BASELINE.json's metric ("... + FER@Eb/N0 sweep"): `frames` frames per point through the
    device-side transmit chain + decoder + compare (polar_fer_batch), outside the timed region, next to the BLER the
    reference's own logs hold for this configuration (myResult_1024/CASCL_L8.dat: block errors / run, all seeds)."""
    pub = {}
    try:
        with open(os.path.join(REPO, "tests", "golden", "published_runs.json")) as f:
            for block in json.load(f).get("myResult_1024/CASCL_L8.dat", []):
                for snr, ble, run in block["rows"]:
                    e, r = pub.get(snr, (0, 0))
                    pub[snr] = (e + ble, r + run)
    except (OSError, ValueError, KeyError):
        pub = {}
    rows = []
    for snr in (1.0, 1.5, 2.0, 2.5, 3.0):
        blk, bits = dec.fer_batch(1242, 0, snr, frames)
        row = {"snr_db": snr, "frames": frames, "block_errors": blk, "fer": blk / float(frames)}
        if snr in pub:
            row["published_bler"] = pub[snr][0] / float(pub[snr][1])
        rows.append(row)
    return rows


def other_configs(pa, torch, device, local, snr_db):
    """BASELINE.json's other configurations on this GPU, f64, at BASELINE's batch sizes: one launch each, timed with HIP
    events on the decoder's stream (polar_time_decode_device), LLRs resident in HBM.  Outside the timed region; the
    headline `value` is not affected.  (tools/bench_configs.py is the developer form of this table.)"""
    sigma = 10 ** (-snr_db / 20)
    rows = []
    for name, mk, N, B in (
            ("BP_1024: N=1024 K=512 BP 50 iterations, 2^16 frames", lambda: pa.BP(1024, 512, iterMax=50, device=local), 1024, 1 << 16),
            ("SCL_1024: N=1024 K=512 SCL L=8, 2^16 frames", lambda: pa.SCLdecode(1024, 512, L=8, device=local), 1024, 1 << 16),
            ("N=4096 K=2048 CA-SCL L=32 (LLR levels spill HBM), 2^15 frames per GPU",
             lambda: pa.CASCL(4096, 2048, L=32, device=local), 4096, 1 << 15),
            ("SC_1024: N=1024 K=512 SC, 2^18 frames", lambda: pa.SCdecode(1024, 512, device=local), 1024, 1 << 18),
            ("BP_128 (not a BASELINE config; the reference's BP_128.c): N=128 K=64 BP 100 iterations, 2^18 frames",
             lambda: pa.BP(128, 64, iterMax=100, device=local), 128, 1 << 18)):
        try:
            d = mk()
            y = 1.0 + sigma * torch.randn(B, N, dtype=torch.float64, device=device)
            x = (2 * y / sigma / sigma).contiguous()
            del y
            ob = torch.empty((B, N // 32), dtype=torch.int32, device=device)
            torch.cuda.synchronize()
            d.decode_device(x, out_bits=ob)
            d.synchronize()
            ms = d.time_decode_device(x, ob, 2)
            rows.append({"config": name, "kernel": d.kernel_name, "frames": B, "kernel_ms": ms, "frames_per_s": B / ms * 1e3,
                         "frames_in_error": int((ob != 0).any(dim=1).sum().item())})
            del d, x, ob
        except Exception as e:  # pragma: no cover
            rows.append({"config": name, "error": str(e)})
    return rows


def cpu_baseline(snr_db, seconds_target=15.0):
    """The reference's CASCL() itself (oracle/_ref/libCASCL_1024_L8.so, built from /root/reference by
    oracle/Makefile), one thread, decode call only, on frames of the same distribution."""
    import numpy as np
    try:
        from oracle import oracle_py as O
    except Exception as e:  # pragma: no cover
        return {"value": None, "unit": "frames/s", "cores": 1, "kind": "port", "sample": f"oracle unavailable: {e}"}
    cpu = f"{host_cpu_model()}, {os.cpu_count()} logical cores present, 1 used"
    code = O.Code(N, K, O.CRC24C_TAPS)
    sig = O.sigma_from_db(snr_db)
    sim = O.Sim(1024)
    if O.ref_available("CASCL_1024_L8"):
        ref = O.Ref("CASCL_1024_L8")
        nfr = 256
        us, ys = sim.frames(code, sig, nfr)
        t = ref.time_decode(ys[:32], sig)  # probe
        rate = 32 / t
        reps = max(1, int(seconds_target * rate / nfr))
        tot = 0.0
        for _ in range(reps):
            tot += ref.time_decode(ys, sig)
        return {"value": reps * nfr / tot, "unit": "frames/s", "cores": 1, "kind": "reference",
                "sample": f"{reps * nfr} frames @ {snr_db} dB, CASCL() of CASCL_1024_L8.c compiled gcc -O2 (oracle/_ref, "
                          f"untracked build product of oracle/Makefile), decode call only; host: {cpu}"}
    nfr = 512
    us, ys = sim.frames(code, sig, nfr)
    llr = np.stack([O.llr_from_y(y, sig) for y in ys])
    t0 = time.perf_counter()
    cnt = 0
    while time.perf_counter() - t0 < seconds_target:
        O.lib().po_decode_batch_f64(code._h, 3, 8, 0, llr.ctypes.data_as(O.C.POINTER(O.C.c_double)), nfr, None)
        cnt += nfr
    return {"value": cnt / (time.perf_counter() - t0), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{cnt} frames @ {snr_db} dB, build's C restatement (oracle/polar_oracle.c), single thread; host: {cpu}"}


def lane_ops_per_frame(N, K, L, r):
    """SURVEY 8d's algorithmic VALU work per frame, (low, high): L (N/2) n CHK at 25 / 40 lane-ops, as many g at 1,
    N L PHI at 8, one rank-by-counting of the 2L candidates (2L x 2L compares) per unfrozen leaf."""
    n = N.bit_length() - 1
    chk = L * (N // 2) * n
    rest = chk + N * L * 8 + (K + r) * (2 * L) * (2 * L)
    return chk * 25 + rest, chk * 40 + rest


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cascl_1024_l8",
                    help="workload: BASELINE.json config 4 (default, the headline metric) or config 5")
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU per step (default: the config's BASELINE batch)")
    ap.add_argument("--snr", type=float, default=2.0)
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fer-sweep", action="store_true", help="skip the FER-vs-Eb/N0 points (outside the timed region)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the kernel-timed rates of BASELINE.json's other configurations (outside the timed region)")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the host-buffer figures (polar_decode_batch from pageable memory, polar_decode latency)")
    ap.add_argument("--one-stream", action="store_true", help="all steps on one stream (no overlap of consecutive launches)")
    ap.add_argument("--streams", type=int, default=2, help="contexts / HIP streams the steps alternate over")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="no GPU: launcher, rendezvous (gloo), barriers, reductions and the JSON relay only; nothing is "
                         "decoded and `value` is null (the world-size-2 CPU test of the N-rank path)")
    return ap.parse_args(argv)


def launch(args, argv):
    """--gpus N > 1 without a launcher: be the launcher.  This process makes no GPU call (it has not even imported
    torch); the ranks are children of torch.distributed.run, rank 0 prints the JSON line, which passes through."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this image
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch(args, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    run(args)


def gather_names(dist, world, name, torch, device):
    """Every rank's device name, through the same backend as the counters (fixed-size byte tensors)."""
    buf = torch.zeros(96, dtype=torch.uint8, device=device)
    raw = name.encode()[:96]
    buf[:len(raw)] = torch.tensor(list(raw), dtype=torch.uint8)
    if not dist:
        return [name]
    allb = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(allb, buf)
    return [bytes(t.cpu().tolist()).rstrip(b"\0").decode(errors="replace") for t in allb]


def host_api_figures(pa, np, local, snr_db, B=1 << 17):
    """What the drop-in costs through the reference's own call shapes (outside the timed region, rank 0, config 4):
    end_to_end = polar_decode_batch from a pageable host double[B][N] to a host int[B][N] (PCIe both ways, chunked pipeline);
    single_frame_latency_us = median of 1000 polar_decode(ctx, y, sigma, u_hat) calls, the literal replacement of
    `CASCL(y, u_hat)` at CASCL_1024_L8.c:294 / SCL_1024.c:263."""
    dec = pa.CASCL(N, K, L=8, crc_taps=CRC, device=local)
    sigma = 10 ** (-snr_db / 20)
    rng = np.random.default_rng(7)
    llr = np.empty((B, N), dtype=np.float64)
    for i in range(0, B, 8192):   # all-zero codeword + noise: valid for every linear code
        llr[i:i + 8192] = 2.0 * (1.0 + sigma * rng.standard_normal((min(8192, B - i), N))) / sigma / sigma
    out = np.empty((B, N), dtype=np.int32)
    dec.decode_batch(llr[:16384], out=out[:16384])          # buffers, pinned staging, first touch of `out`
    out[:] = 0
    t = []
    for _ in range(2):
        t0 = time.perf_counter()
        dec.decode_batch(llr, out=out)
        t.append(time.perf_counter() - t0)
    sec = min(t)
    moved = B * N * 8 + B * (N // 8)       # over PCIe: LLRs in, packed decisions out (unpacked to int[N] on the host)
    y = 1.0 + sigma * rng.standard_normal(N)
    lat = []
    for _ in range(1000):
        t0 = time.perf_counter()
        dec(y, sigma)
        lat.append(time.perf_counter() - t0)
    lat.sort()
    return ({"value": B / sec, "unit": "frames/s", "frames": B, "seconds": sec, "pcie_bytes": moved,
             "pcie_GBps": moved / sec / 1e9, "host_bytes_out": B * N * 4,
             "path": "polar_decode_batch: pageable double llr[B][N] -> int u_hat[B][N] on the host (caller reuses its "
                     "output array as the reference does), 16384-frame chunks, pinned staging, copy / decode / unpack overlapped",
             "frames_in_error": int((out != 0).any(axis=1).sum())},
            {"median": lat[500] * 1e6, "p10": lat[100] * 1e6, "p90": lat[900] * 1e6, "calls": 1000,
             "call": "polar_decode(ctx, y, sigma, u_hat): one frame, host pointers, synchronous"})


def run(args):
    # RCCL on this image needs dmabuf IPC; the variable must be in the environment before the HIP runtime starts, also when
    # an external torchrun (not launch() above) started this rank.  Environment only -- nothing is re-executed.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import polardecoding_amd as pa
    from polardecoding_amd.synth import make_batch

    cfg = CONFIGS[args.config]
    Nc, Kc, Lc = cfg["N"], cfg["K"], cfg["L"]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:   # launched by torch.distributed.run: RCCL even for one rank
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_cpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    if args.rehearse_cpu:
        return rehearse(args, dist, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    dtype = pa.F64 if args.dtype == "f64" else pa.F32
    # Two contexts, each with its own HIP stream, scratch and output buffer: step i runs on context i & 1, so the
    # last, partly filled pass of one launch (131072 frames = 21.3 passes of the resident wavefronts) overlaps the
    # first pass of the next instead of leaving CUs idle.  Every step is still one full decode of one batch.
    decs = [pa.CASCL(Nc, Kc, L=Lc, crc_taps=CRC, dtype=dtype, device=local) for _ in range(1 if args.one_stream else args.streams)]
    dec = decs[0]
    info = torch.tensor(dec.info_order, device=device, dtype=torch.long)   # I[]: 5G order (N <= 1024), beta expansion above

    gen = torch.Generator(device=device)
    gen.manual_seed(20261004 + rank)
    B = args.batch or cfg["batch"]
    in_dtype = torch.float64 if args.dtype == "f64" else torch.float32
    batches = [make_batch(B, Nc, Kc, CRC, args.snr, info, device, gen, in_dtype) for _ in range(2)]
    outs = [torch.empty((B, Nc // 32), dtype=torch.int32, device=device) for _ in decs]
    out_bits = outs[0]
    counters = torch.zeros(2, dtype=torch.int64, device=device)
    torch.cuda.synchronize()   # the batches were made on torch's stream; the decoders run on their own

    def step(i):
        llr, _ = batches[i & 1]
        decs[i % len(decs)].decode_device(llr, out_bits=outs[i % len(decs)])

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    # FER of the last step (outside the timed region): device compare + RCCL sum of two counters
    last = args.steps - 1
    decs[last % len(decs)].count_errors_device(outs[last % len(decs)], batches[last & 1][1], counters)
    torch.cuda.synchronize()
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    per_rank = [elapsed]
    ones = torch.ones(1, dtype=torch.int64, device=device)
    if dist:
        tall = [torch.zeros_like(tmax) for _ in range(world)]
        dist.all_gather(tall, tmax)                   # every rank's own time: stragglers show in the 1 -> 8 curve
        per_rank = [float(t.item()) for t in tall]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)   # how many ranks the collective backend really summed over
    rccl_ranks = int(ones.item())
    dev_names = gather_names(dist, world, torch.cuda.get_device_name(local), torch, device)
    elapsed = float(tmax.item())
    blk, bits = [int(v) for v in counters.tolist()]

    # dominant-kernel timing with HIP events on the kernel's own stream (rank 0's GPU)
    # (one untimed launch first: the timed region above alternated over two contexts, and the first launch alone on this
    # context's scratch after that is 5-10 % slower than the ones that follow it -- r02 / r03 kernel traces)
    dec.decode_device(batches[0][0], out_bits=out_bits)
    dec.synchronize()
    reps = max(3, min(10, args.steps))
    ms_kernel = dec.time_decode_device(batches[0][0], out_bits, reps)
    in_bytes = 8 if args.dtype == "f64" else 4
    alg_bytes = B * (Nc * in_bytes + Nc // 8)  # LLRs in, packed bits out
    achieved = alg_bytes / (ms_kernel * 1e-3) / 1e9
    single_fps = B / ms_kernel * 1e3

    # HBM traffic and VALU counters cannot be read from inside this process: they come from separate rocprofv3 --pmc
    # passes over this same command (tools/prof_pmc.sh), whose summary is committed as profiles/<round>_traffic.json
    # together with the commit it was measured at.  FETCH_SIZE is corrected by the factor calibrated on gfx950
    # (profiles/r03_counter_calibration.txt: reads are reported at half their bytes, writes exactly); the raw sum is kept
    # beside it.  Reported only for the profiled workload, and labelled as such.
    traffic = traffic_raw = None
    traffic_source = None
    valu_prof = {}
    try:
        with open(os.path.join(REPO, "profiles", TRAFFIC_PROFILE)) as f:
            tall_ = json.load(f)
        tj = tall_.get(cfg["traffic_key"] or args.dtype)
        if tj and tj["frames_per_launch"] == B and (cfg["traffic_key"] is None or args.dtype == "f64"):
            traffic_raw = (tj["fetch_kib_raw"] + tj["write_kib_raw"]) * 1024.0
            traffic = (tj["fetch_kib_raw"] * tall_["fetch_factor"] + tj["write_kib_raw"] * tall_["write_factor"]) * 1024.0
            traffic_source = (f"from_profile: profiles/{TRAFFIC_PROFILE} (rocprofv3 --pmc, commit {tall_.get('commit', '?')}), not "
                              f"measured in this run; corrected = FETCH_SIZE x {tall_['fetch_factor']:g} + WRITE_SIZE x "
                              f"{tall_['write_factor']:g} (profiles/r03_counter_calibration.txt)")
            valu_prof = {"busy_frac": tj.get("valu_busy_frac"), "insts_per_frame": tj.get("valu_insts_per_frame"),
                         "source": traffic_source}
    except Exception:
        traffic = traffic_raw = None
    lo, hi = lane_ops_per_frame(Nc, Kc, Lc, R)
    valu = dict(valu_prof)
    valu.update({"peak_lane_ops_per_s": VALU_PEAK_LANE_OPS, "lane_ops_per_frame": [lo, hi],
                 "achieved_frac": [single_fps * lo / VALU_PEAK_LANE_OPS, single_fps * hi / VALU_PEAK_LANE_OPS],
                 "definition": "SURVEY 8d: single-launch frames/s x algorithmic lane-ops per frame / 3.9e13; lane-ops = "
                               "L (N/2) n CHK at 25 (low) / 40 (high) + as many g + 8 N L for PHI + (2L)^2 compares per "
                               "unfrozen leaf; busy_frac / insts_per_frame (profile) are what the kernel actually issues"})

    # secondary figure (not `value`): the f32 instantiation of the same kernel on the same batch, kernel-timed
    secondary = None
    if rank == 0 and args.dtype == "f64":
        try:
            dec32 = pa.CASCL(Nc, Kc, L=Lc, crc_taps=CRC, dtype=pa.F32, device=local)
            dec32.use_torch_stream()
            llr32 = batches[0][0].float().contiguous()
            dec32.decode_device(llr32, out_bits=out_bits)
            torch.cuda.synchronize()
            ms32 = dec32.time_decode_device(llr32, out_bits, 3)
            secondary = {"dtype": "f32", "frames_per_s_one_gpu": B / ms32 * 1e3, "kernel": dec32.kernel_name,
                         "note": "same operation order in binary32: FER-equivalent, not bit-identical to the reference"}
            del dec32, llr32
        except Exception as e:  # pragma: no cover
            secondary = {"error": str(e)}

    headline = args.config == "cascl_1024_l8"
    fer_sweep = None
    if rank == 0 and headline and not args.no_fer_sweep:
        fer_sweep = fer_vs_snr(dec, B)
    other = None
    if rank == 0 and world == 1 and headline and not args.no_other_configs:
        other = other_configs(pa, torch, device, local, args.snr)
    e2e = lat = None
    if rank == 0 and world == 1 and headline and not args.no_end_to_end:
        try:
            e2e, lat = host_api_figures(pa, np, local, args.snr)
        except Exception as e:  # pragma: no cover
            e2e = {"error": str(e)}

    if rank == 0:
        total_frames = world * B * args.steps
        value = total_frames / elapsed
        base = BASELINE_MD_CPU.get(args.config)
        out = {
            "metric": cfg["metric"],
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "per_rank_frames_per_s": [B * args.steps / t for t in per_rank],
            "single_launch_frames_per_s": single_fps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": (value / base) if base else None,
            "vs_baseline_source": (f"BASELINE.md section 2: reference {args.config} C, one core, decode call only, {base} frames/s "
                                   "(survey-measured: the reference publishes no throughput); the same program timed on "
                                   "THIS host is cpu_baseline") if base else None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{cfg['label']}, {B} frames/GPU/step ({cfg['over8']}), BPSK-AWGN Eb/N0={args.snr} dB, "
                                   "LLRs resident in HBM",
                       "name": args.config, "frames_per_gpu_per_step": B, "snr_db": args.snr,
                       "parallelism": f"frames sharded x{world}", "streams_per_gpu": len(decs)},
            "rccl_ranks": rccl_ranks,
            "devices": dev_names,
            "fer": {"block_errors": blk, "bit_errors": bits, "frames": world * B,
                    "fer": blk / float(world * B)},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_raw": traffic_raw,
                         "traffic_source": traffic_source,
                         "kernel": dec.kernel_name, "kernel_ms": ms_kernel,
                         "algorithmic_bytes_per_launch": alg_bytes, "valu": valu,
                         "note": "achieved / peak / frac are the contract's HBM figures (algorithmic bytes per launch / "
                                 "kernel_ms against 8 TB/s): small by construction, this path is VALU-issue bound, not "
                                 "HBM-bound (SURVEY.md 0.5) -- the binding roofline is `valu` (achieved_frac).  kernel_ms is "
                                 "one launch alone on its stream; with two streams consecutive steps overlap their partly "
                                 "filled last pass, so ms_per_step can be below kernel_ms"},
        }
        if fer_sweep:
            out["fer_sweep"] = fer_sweep
        if other:
            out["other_configs"] = other
        if secondary:
            out["secondary"] = secondary
        if e2e:
            out["end_to_end"] = e2e
        if lat:
            out["single_frame_latency_us"] = lat
        if world == 1 and headline and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.snr)
            if out["cpu_baseline"].get("value"):
                out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()   # rank 0 still measures the kernel-level figures after the timed region: leave together
        dist.destroy_process_group()


def rehearse(args, dist, rank, world):
    """--rehearse-cpu: the N-rank control path without a GPU -- same barriers, same reductions (gloo instead of RCCL),
    same JSON relay; the step is empty and `value` is null.  Nothing here is a measurement."""
    import torch
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64)
    counters = torch.tensor([rank + 1, 10 * (rank + 1)], dtype=torch.int64)   # stand-ins for (block, bit) errors
    per_rank = [elapsed]
    ones = torch.ones(1, dtype=torch.int64)
    if dist:
        tall = [torch.zeros_like(tmax) for _ in range(world)]
        dist.all_gather(tall, tmax)
        per_rank = [float(t.item()) for t in tall]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
    names = gather_names(dist, world, f"cpu (rehearsal, rank {rank})", torch, torch.device("cpu"))
    cfg = CONFIGS[args.config]
    if rank == 0:
        print(json.dumps({"metric": cfg["metric"], "value": None, "unit": "frames/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "rehearsal": True,
                          "config": {"name": args.config, "frames_per_gpu_per_step": args.batch or cfg["batch"],
                                     "workload": cfg["label"]},
                          "rccl_ranks": int(ones.item()), "devices": names,
                          "ranks_seen": len(per_rank), "counter_sum": [int(v) for v in counters.tolist()],
                          "note": "--rehearse-cpu: launcher / rendezvous / reductions only, nothing decoded"}), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
