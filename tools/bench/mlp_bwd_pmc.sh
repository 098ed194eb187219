cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mlp_bwd_pmc; mkdir -p $O
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/$i -o p -- python3 $R/bench.py --no-conv --no-cpu-baseline --steps 2 --warmup 1 --adjoint-steps 2 > $O/run$i.log 2>&1 || { tail -3 $O/run$i.log; exit 1; }
done
python3 - <<PY
import csv,re,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1,2):
    for r in csv.DictReader(open("$O/%d/p_counter_collection.csv"%i)):
        m=re.search(r"(k_\w+(<[^>]*>)?)",r["Kernel_Name"])
        if m: acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print("  %-22s n=%4d  FETCH %9.1f KiB  WRITE %9.1f KiB"%(k,len(v.get("FETCH_SIZE",[])),sum(v.get("FETCH_SIZE",[0]))/max(len(v.get("FETCH_SIZE",[1])),1),sum(v.get("WRITE_SIZE",[0]))/max(len(v.get("WRITE_SIZE",[1])),1)))
PY
