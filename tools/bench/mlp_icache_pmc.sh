cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/icache_pmc; mkdir -p $O
i=0
for ctrs in "SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQC_ICACHE_REQ SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_BUSY_CYCLES" "SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/$i -o p -- python3 $R/bench.py --no-conv --no-cpu-baseline --steps 3 --warmup 1 --adjoint-steps 1 > $O/run$i.log 2>&1 || { tail -3 $O/run$i.log; exit 1; }
done
python3 - <<PY
import csv,re,collections,glob
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m=re.search(r"(k_\w+(<[^>]*>)?)",r["Kernel_Name"])
        if m and m.group(1) in ("k_step_q<false, 1>","k_vjp_q_pg<1>"): acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k)
    for c in sorted(v):
        x=v[c]; big=[y for y in x if y>0.2*max(x)] if max(x)>0 else x
        print("   %-28s %14.0f"%(c,sum(big)/max(len(big),1)))
PY
