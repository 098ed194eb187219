# per-kernel breakdown of the MLP bench incl. training steps: rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mlpprof -o mlp -- python3 $R/bench.py --no-conv --no-cpu-baseline --steps 3 --warmup 1 --adjoint-steps 10 > $R/gpurun_out/mlpprof.log 2>&1 || exit 1
python3 - <<PY
import csv,re
for r in csv.DictReader(open("$R/gpurun_out/mlpprof/mlp_kernel_stats.csv")):
    m=re.search(r"(k_\w+(<[^>]*>)?)",r["Name"])
    n=m.group(1) if m else r["Name"][:40]
    print("  %-40s n=%5s avg=%8.1f us tot=%8.2f ms"%(n,r["Calls"],float(r["AverageNs"])/1e3,float(r["TotalDurationNs"])/1e6))
PY
