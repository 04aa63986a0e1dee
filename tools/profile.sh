#!/bin/bash
# Profiling recipe (run on the GPU box via gpurun): kernel trace + stats, then PMC passes, each in its own run.
# Usage: [PASSES="trace sq1 sq2 tcc fetch write tcp"] bash tools/profile.sh <tag> [bench args...]
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
python3 -c "import sys; sys.path.insert(0, '$R'); from rustracer_amd._lib import kernel_source_hash as h; print(h())" > $R/$OUT/source_hash.txt
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --cpu-seconds 0 --stats-spp 0 $@"
# the same workload once without the profiler but with the instrumented pass: its line carries the units (light samples, traced rays, intersect
# rays) each kernel processes per launch, which turn the per-launch counters below into per-unit figures
timeout -k 10 300 python3 $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --stats-spp 2 "$@" > $R/$OUT/units.log 2>&1 || { tail -5 $R/$OUT/units.log; exit 1; }
PASSES=${PASSES:-"trace sq1 sq2 tcc fetch write tcp"}
declare -A PMC
PMC[sq1]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
PMC[sq2]="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM"
PMC[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
PMC[fetch]="FETCH_SIZE"
PMC[write]="WRITE_SIZE"
PMC[tcp]="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"
for P in $PASSES; do
  if [ "$P" = trace ]; then
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace -- python3 $R/bench.py $ARGS > $R/$OUT/trace.log 2>&1 || { tail -5 $R/$OUT/trace.log; exit 1; }
  else
    timeout -s KILL 240 rocprofv3 --pmc ${PMC[$P]} --output-format csv -d $R/$OUT/pmc_$P -- python3 $R/bench.py $ARGS > $R/$OUT/pmc_$P.log 2>&1 || { echo "pass $P failed"; tail -3 $R/$OUT/pmc_$P.log; exit 1; }
  fi
done
echo profiled $TAG
