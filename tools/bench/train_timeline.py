# Kernel timeline of the LAST training step (from its parameter repack k_pack to the end of the trace) in a rocprofv3
# --kernel-trace CSV, both queues, runs of the same kernel on the same queue condensed:
#   python tools/bench/train_timeline.py <kernel_trace.csv>
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_pack" in r["Kernel_Name"]]
if not idx: sys.exit("no k_pack in the trace")
seg = rows[idx[-1]:]
qs = sorted(set(r.get("Queue_Id", "0") for r in seg))
def short(n):
    m = re.search(r"(k_[A-Za-z0-9_]+(<[^>(]*>)?)", n)
    return (m.group(1) if m else n.split("::")[-1])[:40]
t0 = int(seg[0]["Start_Timestamp"]); prev_end = {}; runs = []
for r in seg:
    s, e, q, n = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), qs.index(r.get("Queue_Id", "0")), short(r["Kernel_Name"])
    gap = (s - prev_end.get(q, s)) / 1e3
    prev_end[q] = e
    if runs and runs[-1]["n"] == n and runs[-1]["q"] == q and gap < 1.0:
        runs[-1]["cnt"] += 1; runs[-1]["dur"] += (e - s) / 1e3; runs[-1]["end"] = e
    else:
        runs.append(dict(n=n, q=q, s=s, end=e, cnt=1, dur=(e - s) / 1e3, gap=gap))
for u in runs:
    print(f"{(u['s'] - t0) / 1e3:9.1f} us  q{u['q']}  gap {u['gap']:7.2f}  {u['cnt']:3d} x {u['dur'] / u['cnt']:7.2f} us = {u['dur']:8.1f}  {u['n']}")
last = max(int(r["End_Timestamp"]) for r in seg)
busy = {}
for r in seg: busy[qs.index(r.get("Queue_Id", "0"))] = busy.get(qs.index(r.get("Queue_Id", "0")), 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(f"span {(last - t0) / 1e3:.1f} us, busy " + ", ".join(f"q{q} {b / 1e3:.1f}" for q, b in sorted(busy.items())))
