# traversal statistics on the GPU box: bash tools/stats.sh "opt=val ..." ...
set -o pipefail
for o in "$@"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  echo "== $o"
  timeout -k 10 200 python -u bench.py --spp ${SPP:-16} --steps 1 --warmup 1 --cpu-seconds 0 --stats-spp 2 $args 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['kernel_ms_last_step']); print(r['per_traced_ray'], r['bytes_per_ray'])" || exit 1
done
