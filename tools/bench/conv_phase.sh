cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in ${DBGS:-0 1 4 5 2 6 7}; do
  export LRNDE_CONV_DBG=$d
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ph$d -o ph -- python3 $R/tools/bench/conv_bf16_phase.py ${DT:-bf16} > $R/gpurun_out/ph$d.log 2>&1 || exit 1
  echo "== dbg=$d"; tail -1 $R/gpurun_out/ph$d.log
  python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/ph$d/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    import re
    m=re.search(r"(k_\w+(<[^>]*>)?)",r["Name"])
    if m and ("conv" in m.group(1) or "bn_fin" in m.group(1)):
        print("  %-40s n=%s avg=%.1f us"%(m.group(1),r["Calls"],float(r["AverageNs"])/1e3))
PY
done
