# Condensed kernel timeline of the last <ms> milliseconds of a rocprofv3 --kernel-trace CSV (all queues):
#   python tools/bench/tail_timeline.py <kernel_trace.csv> [ms=20]
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
end = max(int(r["End_Timestamp"]) for r in rows)
seg = [r for r in rows if int(r["Start_Timestamp"]) >= end - ms * 1e6]
qs = sorted(set(r.get("Queue_Id", "0") for r in seg))
def short(n):
    m = re.search(r"(k_[A-Za-z0-9_]+(<[^>(]*>)?)", n)
    return (m.group(1) if m else n.split("::")[-1])[:44]
t0 = int(seg[0]["Start_Timestamp"]); prev_end = {}; runs = []; gaps = {}
for r in seg:
    s, e, q, n = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), qs.index(r.get("Queue_Id", "0")), short(r["Kernel_Name"])
    gap = (s - prev_end.get(q, s)) / 1e3
    prev_end[q] = e
    if gap > 1.0: gaps[q] = gaps.get(q, 0.0) + gap
    if runs and runs[-1]["n"] == n and runs[-1]["q"] == q and gap < 1.0:
        runs[-1]["cnt"] += 1; runs[-1]["dur"] += (e - s) / 1e3
    else:
        runs.append(dict(n=n, q=q, s=s, cnt=1, dur=(e - s) / 1e3, gap=gap))
for u in runs:
    print(f"{(u['s'] - t0) / 1e3:10.1f} us  q{u['q']}  gap {u['gap']:8.2f}  {u['cnt']:3d} x {u['dur'] / u['cnt']:8.2f} us = {u['dur']:9.1f}  {u['n']}")
busy = {}
for r in seg:
    q = qs.index(r.get("Queue_Id", "0")); busy[q] = busy.get(q, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(f"span {(end - t0) / 1e3:.1f} us; " + ", ".join(f"q{q}: busy {b / 1e3:.1f}, gaps > 1 us {gaps.get(q, 0.0):.1f}" for q, b in sorted(busy.items())))
