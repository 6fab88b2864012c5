#!/usr/bin/env python3
"""Summarise a tools/prof_pmc.sh output directory into a small text file for profiles/.
usage: summarize_prof.py gpurun_out/profN frames_per_launch > profiles/rNN_xxx.txt"""
import collections, csv, glob, sys
d, frames = sys.argv[1], int(sys.argv[2])
# gfx950 calibration (profiles/r03_counter_calibration.txt, tools/calib_traffic.hip + tools/calibrate_counters.py): on known
# byte counts FETCH_SIZE reports exactly half of the bytes read for every access shape the decoders use (4 / 8 / 16 B per
# lane, plain and sc1, 32-byte pieces, scratch (spill) reloads); WRITE_SIZE reports the bytes written exactly.  Raw values
# are printed as rocprofv3 gives them, and corrected = raw x factor beside them.
CORRECTION = {"FETCH_SIZE": 2.0, "WRITE_SIZE": 1.0}
print(f"# rocprofv3 summary of {d} (frames per launch = {frames})")
for f in glob.glob(f"{d}/stats/*/*_kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    print("## kernel-trace --stats (top kernels)")
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage")
    for r in rows[:4]:
        print(",".join([r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]]))
# per-dispatch view of the decode kernels: bench.py alternates its steps over two streams, so consecutive dispatches
# time-share the CUs and the trace stretches them; the dispatches that ran alone are the kernel's own duration
for f in glob.glob(f"{d}/stats/*/*_kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "polar::k_" in r["Kernel_Name"] and "k_count" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if rows:
        print("## kernel-trace, per dispatch (ms from the first; 'alone' = overlaps no other decode dispatch)")
        t0 = int(rows[0]["Start_Timestamp"])
        alone = collections.defaultdict(list)
        for i, r in enumerate(rows):
            a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            # back-to-back dispatches on one stream touch by a few hundred ns: only an overlap of more than 1 % counts
            tol = (b - a) // 100
            ov = any(j != i and min(b, int(q["End_Timestamp"])) - max(a, int(q["Start_Timestamp"])) > tol for j, q in enumerate(rows))
            kn = r["Kernel_Name"].split("(")[0].replace("void polar::", "")
            print(f"{kn:42s} start {(a - t0) / 1e6:9.3f}  duration {(b - a) / 1e6:8.3f} ms  {'overlapped' if ov else 'alone'}")
            if not ov:
                alone[kn].append(((b - a) / 1e6, i))
        first = {}
        for i, r in enumerate(rows):
            first.setdefault(r["Kernel_Name"].split("(")[0].replace("void polar::", ""), i)
        for kn, v in alone.items():
            print(f"{kn:42s} average of the {len(v)} dispatches that ran alone: {sum(x for x, _ in v) / len(v):.3f} ms")
            w = [x for x, i in v if i != first[kn]]   # the kernel's first dispatch of the process is the cold warm-up step
            if w and len(w) != len(v):
                print(f"{kn:42s} ... without the kernel's first (cold) dispatch: {sum(w) / len(w):.3f} ms over {len(w)}")
print("## PMC (sum over the chip, per dispatch of the decode kernel, and per frame)")
for sub in sorted(glob.glob(f"{d}/pmc*")):
    for f in glob.glob(f"{sub}/*/*_counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in rows:
            kn = r["Kernel_Name"]
            if "polar::k_" in kn and "k_count" not in kn and "k_generate" not in kn:
                kn = kn.split("(")[0].replace("void polar::", "")
                agg[kn][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[kn].add(r["Dispatch_Id"])
        for kn in sorted(agg):
            for c, v in sorted(agg[kn].items()):
                per = v / max(1, len(disp[kn]))
                print(f"{kn:42s} {c:22s} {per:14.6g} per dispatch {per / frames:12.3f} per frame ({len(disp[kn])} dispatches)")
                if c in CORRECTION:
                    k = CORRECTION[c]
                    print(f"{kn:42s} {c + ' corrected':22s} {per * k:14.6g} KiB per dispatch {per * k / frames:12.3f} KiB per frame (raw x {k:g}, gfx950 calibration)")
