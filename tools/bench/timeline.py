# Prints the kernel timeline (start offset, duration, gap to the previous kernel) of the LAST backward pass in a rocprofv3
# --kernel-trace output (sqlite database or *_kernel_trace.csv):  python tools/bench/timeline.py <file> [first-kernel-substring]
import sqlite3, sys, csv
if sys.argv[1].endswith(".csv"):  # rocprofv3 --output-format csv: *_kernel_trace.csv
    rows = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1]))),
                  key=lambda r: r[1])
else:
    db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kt = [t for t in tabs if 'kernel_dispatch' in t][0]; sym = [t for t in tabs if 'kernel_symbol' in t][0]
    rows = cur.execute(f"select s.kernel_name, k.start, k.end from {kt} k join {sym} s on k.kernel_id=s.id order by k.start").fetchall()
mark = sys.argv[2] if len(sys.argv) > 2 else "k_adj_ctrl_init"
idx = [i for i, r in enumerate(rows) if mark in r[0]]
if not idx: sys.exit("marker kernel not found")
i0 = idx[-1]
# end: the next k_cls / k_pack kernel (next training step) or the end of the trace
i1 = len(rows)
for i in range(i0 + 1, len(rows)):
    if "k_pack" in rows[i][0] or "k_cls_fwd" in rows[i][0]: i1 = i; break
seg = rows[i0:i1]
t0 = seg[0][1]; prev_end = t0; busy = 0; gaps = []
import re
def short(n):
    m = re.search(r"(k_[A-Za-z0-9_]+(<[^>(]*>)?)", n)
    return (m.group(1) if m else n.split("N_1")[-1])[:44]
for n, s, e in seg:
    gaps.append((s - prev_end) / 1e3); busy += (e - s)
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:7.2f}  {short(n)}")
    prev_end = e
span = (seg[-1][2] - t0) / 1e3
print(f"launches {len(seg)}  span {span:.1f} us  busy {busy / 1e3:.1f} us  idle {span - busy / 1e3:.1f} us  (gaps > 5 us: {sum(1 for g in gaps if g > 5)}, sum {sum(g for g in gaps if g > 5):.1f} us)")
