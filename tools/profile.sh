#!/bin/bash
# Profiling recipe (run on the GPU box via gpurun): kernel trace + stats, then PMC passes, each in its own run.
# Usage: bash tools_profile.sh <tag> [bench args...]
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--steps 1 --warmup 1 --cpu-seconds 0 --stats-spp 0 $@"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace -- python3 $R/bench.py $ARGS > $R/$OUT/trace.log 2>&1 || exit 1
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  N=$(echo $PASS | cut -d' ' -f1)
  timeout -s KILL 240 rocprofv3 --pmc $PASS --output-format csv -d $R/$OUT/pmc_$N -- python3 $R/bench.py $ARGS > $R/$OUT/pmc_$N.log 2>&1 || { echo "pass $N failed"; tail -3 $R/$OUT/pmc_$N.log; exit 1; }
done
echo profiled
