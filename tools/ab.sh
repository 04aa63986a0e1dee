# A/B bench on the GPU box: bash tools/ab.sh "opt=val opt=val" "opt=val" ...   (each argument is one bench run; first runs the GPU tests unless NOTEST=1)
set -o pipefail
if [ -z "$NOTEST" ]; then
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t_ab.log 2>&1 || { tail -30 gpurun_out/t_ab.log; exit 1; }
tail -2 gpurun_out/t_ab.log
fi
SPP=${SPP:-16}
for o in "$@"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  echo "== $o"
  timeout -k 10 200 python -u bench.py --spp $SPP --steps 2 --warmup 1 --cpu-seconds 0 --stats-spp 0 $args 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_last_step'])" || exit 1
done
