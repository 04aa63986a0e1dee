# A/B of library variants on the GPU box: bash tools/abv.sh <tag> <tag> ...   ("" = the in-tree library)
set -o pipefail
SPP=${SPP:-16}
for t in "$@"; do
  echo "== variant ${t:-default}"
  if [ -n "$t" ]; then export RUSTRACER_HIP_LIB=$GRAFT_REPO_ROOT/rustracer_amd/variants/lib_$t.so; else unset RUSTRACER_HIP_LIB; fi
  timeout -k 10 200 python -u bench.py --spp $SPP --steps 2 --warmup 1 --cpu-seconds 0 --stats-spp 0 $OPTS 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_last_step'])" || exit 1
done
