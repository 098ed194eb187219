# per-kernel breakdown of one conv-field VJP (CIFAR shape): rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bwdprof -o bwd -- python3 $R/tools/bench/conv_bwd_bench.py > $R/gpurun_out/bwdprof.log 2>&1 || exit 1
grep "conv vjp" $R/gpurun_out/bwdprof.log
python3 - <<PY
import csv,re
for r in csv.DictReader(open("$R/gpurun_out/bwdprof/bwd_kernel_stats.csv")):
    m=re.search(r"(k_\w+(<[^>]*>)?)",r["Name"])
    if m: print("  %-44s n=%4s avg=%8.1f us  %5s %%"%(m.group(1),r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
