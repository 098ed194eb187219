# kernel timeline (both queues) of the last backward pass of tools/bench/adjoint_breakdown.py under rocprofv3 --kernel-trace:
#   bash tools/bench/adjoint_timeline.sh <tag>      (environment switches, e.g. LRNDE_ADJ_OVERLAP=1, are passed through)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/adjtl_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python3 $R/tools/bench/adjoint_breakdown.py > $O/run.log 2>&1 || { tail -3 $O/run.log; exit 1; }
F=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv, re
rows = sorted(csv.DictReader(open("$F")), key=lambda r: int(r["Start_Timestamp"]))
# the last backward pass: from the last k_adj_begin to the end
idx = [i for i, r in enumerate(rows) if "k_adj_begin" in r["Kernel_Name"]]
seg = rows[idx[-1]:]
qs = sorted(set(r.get("Queue_Id", "0") for r in seg))
t0 = int(seg[0]["Start_Timestamp"])
def short(n):
    m = re.search(r"(k_[A-Za-z0-9_]+(<[^>(]*>)?)", n)
    return (m.group(1) if m else n)[:34]
out = []
for r in seg[:80]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  q{qs.index(r.get('Queue_Id', '0'))}  dur {(e - s) / 1e3:7.2f}  {short(r['Kernel_Name'])}")
last = max(int(r["End_Timestamp"]) for r in seg)
out.append(f"span of the pass {(last - t0) / 1e3:.1f} us, {len(seg)} launches")
open("$O/timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:70]))
PY
rm -rf $O/prof
