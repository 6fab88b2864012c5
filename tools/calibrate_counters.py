#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE calibration on gfx950: counter value / true bytes per access shape.
usage: calibrate_counters.py OUT  (OUT/fetch, OUT/write = rocprofv3 --pmc passes over tools/calib_traffic,
OUT/true.txt = its stdout).  Prints the table for profiles/r03_counter_calibration.txt and a JSON line of factors
(multiply a raw counter by `factor` to get bytes) that tools/summarize_prof.py applies."""
import collections, csv, glob, json, re, sys

out = sys.argv[1]
true = {}
hdr = []
for line in open(f"{out}/true.txt"):
    if line.startswith("#"):
        hdr.append(line.rstrip())
    m = re.match(r"TRUE (\S+)\s+read_bytes (\d+) write_bytes (\d+)\s+ms ([\d.]+)\s+GB/s (\d+)", line)
    if m:
        true[m.group(1)] = (float(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5)))


def counters(sub, cname):
    agg, disp = collections.defaultdict(float), collections.defaultdict(set)
    for f in glob.glob(f"{out}/{sub}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != cname:
                continue
            kn = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].split("<")[0]
            agg[kn] += float(r["Counter_Value"])
            disp[kn].add(r["Dispatch_Id"])
    return {k: agg[k] / len(disp[k]) for k in agg}


fetch, write = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
print("# FETCH_SIZE / WRITE_SIZE calibration, rocprofv3 --pmc (separate passes), tools/calib_traffic on gfx950")
for h in hdr:
    print(h)
print("# counter values are KiB per dispatch (mean over the dispatches of the kernel); ratio = counter_KiB * 1024 / true bytes")
print(f"{'kernel':28s} {'true read MiB':>14s} {'FETCH_SIZE MiB':>15s} {'ratio':>7s} {'true write MiB':>15s} {'WRITE_SIZE MiB':>15s} {'ratio':>7s} {'GB/s (timed)':>12s}")
factors = {}
for k, (rd, wr, ms, gbs) in true.items():
    f, w = fetch.get(k, float("nan")), write.get(k, float("nan"))
    fr = f * 1024 / rd if rd else float("nan")
    wr_ = w * 1024 / wr if wr else float("nan")
    print(f"{k:28s} {rd / 2**20:14.1f} {f / 1024:15.1f} {fr:7.3f} {wr / 2**20:15.1f} {w / 1024:15.1f} {wr_:7.3f} {gbs:12.0f}")
    factors[k] = {"fetch_ratio": None if fr != fr else round(fr, 4), "write_ratio": None if wr_ != wr_ else round(wr_, 4)}
print("FACTORS " + json.dumps(factors))
