#!/bin/bash
# tools/prof_quick.sh OUTDIR LIBNAME [config] -- a handful of rocprofv3 --pmc passes over ONE library variant's kernel
# (tools/variant.py worker: 1 warm + a few timed launches), program directly after `--`.
OUT=$1; LIB=$2; CFG=${3:-cascl}
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "VALUBusy SALUBusy" "LdsUtil LdsBankConflict" "LdsLatency" "InstrFetchLatency" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_CYCLES GRBM_GUI_ACTIVE"; do
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
    print(f"{kn:40s} {c:28s} {v / len(d):16.6g} per dispatch ({len(d)})")
PY
