# traversal statistics + one-lane kernel times on the GPU box: bash tools/stats.sh "opt=val ..." ...   (each argument is one bench run)
set -o pipefail
for o in "$@"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  echo "== $o"
  timeout -k 10 300 python -u bench.py --spp ${SPP:-16} --steps 2 --warmup 1 --cpu-seconds 0 --stats-spp 2 $args 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels']
print('Mrays/s', d['value'], 'ms/step', d['ms_per_step'], 'lanes-overlapped kernel ms', d['kernel_ms_last_step'])
print('one-lane frame ms', r.get('frame_ms_one_lane'), {n: (v.get('avg_launch_ms'), v.get('launches_one_lane')) for n, v in k.items()}, 'compute', r.get('compute'))
print('trace per_unit', k.get('k_trace_p', {}).get('per_unit'))
print('shaft', k.get('k_light', {}).get('shaft'))
" || exit 1
done
