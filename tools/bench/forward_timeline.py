# Kernel timeline of the LAST layer forward in a rocprofv3 --kernel-trace CSV (start offset, duration, gap to the previous
# kernel's end, queue):  python tools/bench/forward_timeline.py <kernel_trace.csv>
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a forward opens with k_solve_init or, with the folded initialisation, with the first init kernel on the MAIN queue (the
# local step's own init launches run on the companion queue in the middle of the pass)
from collections import Counter
main_q = Counter(r.get("Queue_Id", "0") for r in rows if "k_step" in r["Kernel_Name"]).most_common(1)[0][0]
idx = [i for i, r in enumerate(rows) if "k_solve_init" in r["Kernel_Name"] or
       ("k_init1" in r["Kernel_Name"] and r.get("Queue_Id", "0") == main_q)]
if not idx: sys.exit("no solve start (k_solve_init / k_init1) in the trace")
seg = rows[idx[-1]:]
t0 = int(seg[0]["Start_Timestamp"]); prev_end = {}; last_end = t0
qs = sorted(set(r.get("Queue_Id", "0") for r in seg))
short = lambda n: n.split("::")[-1][:44]
busy = {}
for r in seg:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0")
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    busy[q] = busy.get(q, 0) + (e - s)
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.2f}  gap {gap:7.2f}  q{qs.index(q)}  {short(r['Kernel_Name'])}")
    prev_end[q] = e; last_end = max(last_end, e)
print(f"launches {len(seg)}  span {(last_end - t0) / 1e3:.1f} us  busy per queue " + ", ".join(f"q{qs.index(q)} {busy[q] / 1e3:.1f}" for q in busy))
