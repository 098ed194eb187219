"""Splits the k_step dispatches of a rocprofv3 --kernel-trace CSV into full steps and
prologue-only launches (the terminal footer launch of each solve and speculative launches that
found the solve finished exit after the device prologue, ~4 us)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
full, short = [], []
for r in rows:
    if "k_step" not in r["Kernel_Name"]:
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    (full if d > 20.0 else short).append(d)
print(f"k_step dispatches: {len(full) + len(short)}  full steps: {len(full)} mean {sum(full)/len(full):.1f} us "
      f"(min {min(full):.1f}, max {max(full):.1f})  prologue-only: {len(short)} mean {sum(short)/max(len(short),1):.1f} us")
