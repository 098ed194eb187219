cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sdeprof -o sde -- python3 $R/bench.py --workload mnist_sde --no-cpu-baseline > $R/gpurun_out/sdeprof.log 2>&1 || exit 1
python3 - <<PY
import csv,re
for r in csv.DictReader(open("$R/gpurun_out/sdeprof/sde_kernel_stats.csv")):
    m=re.search(r"(k_\w+(<[^>]*>)?)",r["Name"])
    if m: print("  %-30s n=%5s avg=%8.1f us"%(m.group(1),r["Calls"],float(r["AverageNs"])/1e3))
PY
