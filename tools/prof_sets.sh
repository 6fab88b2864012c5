#!/bin/bash
# tools/prof_sets.sh OUTDIR LIBNAME CONFIG "SET1" "SET2" ... -- one rocprofv3 --pmc pass per counter set over one library
# variant's kernel (tools/variant.py worker), then per-dispatch means.  Program directly after `--`.
OUT=$1; LIB=$2; CFG=$3; shift 3
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/$OUT/p$i -- python3 $R/tools/variant.py ab --worker $LIB --config $CFG --reps 2 > $R/$OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, set()])
for f in glob.glob("$R/$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "polar::k_" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0].replace("void polar::", ""), r["Counter_Name"])
            agg[k][0] += float(r["Counter_Value"]); agg[k][1].add(r["Dispatch_Id"])
for (kn, c), (v, d) in sorted(agg.items()):
    print(f"$LIB {kn:44s} {c:24s} {v / len(d):16.6g} per dispatch ({len(d)})")
PY
