#!/bin/bash
# Tuning sweep on the GPU box: bash tools/sweep.sh "opt=val opt=val" "opt=val" ...   (each argument is one bench run)
SPP=${SPP:-16}
for o in "$@"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  echo "== $o"
  timeout -k 10 200 python -u bench.py --spp $SPP --steps 2 --warmup 1 --cpu-seconds 0 --stats-spp 0 $args 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_last_step'])" || exit 1
done
