#!/bin/bash
# tools/prof_pmc.sh -- rocprofv3 passes for the headline kernel (run on the GPU box via gpurun).
#   kernel-trace/stats pass + separate PMC passes (never combined with other trace domains).
set -e
OUT=${1:-gpurun_out/prof}
ARGS=${2:-"--steps 3 --warmup 1 --no-cpu-baseline --batch 32768"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- python3 $R/bench.py $ARGS > $R/$OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $R/$OUT/pmc1 -- python3 $R/bench.py $ARGS > $R/$OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $R/$OUT/pmc2 -- python3 $R/bench.py $ARGS > $R/$OUT/pmc2.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_WAVE32_INSTS --output-format csv -d $R/$OUT/pmc3 -- python3 $R/bench.py $ARGS > $R/$OUT/pmc3.log 2>&1 || true
echo done
