#!/bin/bash
# [POLAR_PROF_PROG=tools/bench_configs.py] tools/prof_pmc.sh OUTDIR [args] -- rocprofv3 passes for the headline kernel (run on the GPU box via gpurun).
# One kernel-trace/stats pass and separate --pmc passes (never combined with other trace domains).
# The program itself follows `--` (python3 bench.py ...), no env/bash -c hop.
OUT=${1:-gpurun_out/prof}
shift
ARGS=${@:-"--steps 3 --warmup 1 --no-cpu-baseline --no-fer-sweep --no-other-configs --no-end-to-end"}
R=$GRAFT_REPO_ROOT
PROG=${POLAR_PROF_PROG:-bench.py}   # e.g. tools/bench_configs.py with ARGS "--only BP"
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- python3 $R/$PROG $ARGS > $R/$OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $R/$OUT/pmc1 -- python3 $R/$PROG $ARGS > $R/$OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $R/$OUT/pmc2 -- python3 $R/$PROG $ARGS > $R/$OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -- python3 $R/$PROG $ARGS > $R/$OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write -- python3 $R/$PROG $ARGS > $R/$OUT/pmc_write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/$OUT/pmc_l2 -- python3 $R/$PROG $ARGS > $R/$OUT/pmc_l2.log 2>&1
echo done
