cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in 1 32; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$L -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --stats-spp 0 --spp 16 --light-samples $L > $R/gpurun_out/kt_$L.log 2>&1 || exit 1
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/kt_$L/*/*_kernel_stats.csv")[0]
print("L=$L")
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "rt::k_" in n: print("  %-28s calls %4s total %8.2f ms avg %7.3f"%(n.split("rt::")[1][:28], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6))
PY
done
