set -o pipefail
for o in "wide=1" "wide=0"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  echo "== $o"
  timeout -k 10 200 python -u bench.py --spp 16 --steps 1 --warmup 1 --cpu-seconds 0 --stats-spp 2 $args 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'])" || exit 1
done
