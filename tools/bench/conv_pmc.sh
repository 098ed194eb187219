# PMC passes over the conv f-eval kernels: bash tools/bench/conv_pmc.sh <dtype> "<CTR1 CTR2>" "<CTR3>" ...  (one rocprofv3 run per argument)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
dt=$1; shift
i=0
for ctrs in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_$dt/$i -o p -- python3 $R/tools/bench/conv_pmc_run.py $dt > $R/gpurun_out/pmc_$dt.$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$dt.$i.log; exit 1; }
  python3 - <<PY
import csv,re,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open("$R/gpurun_out/pmc_$dt/$i/p_counter_collection.csv")):
    m=re.search(r"(k_\w+(<[^>]*>)?)",r["Kernel_Name"])
    if not m: continue
    acc[m.group(1)][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(m.group(1),r["Counter_Name"])]+=1
for k,v in acc.items():
    print("  %-34s"%k, "  ".join("%s=%.4g"%(c,x/cnt[(k,c)]) for c,x in v.items()))
PY
done
