#!/usr/bin/env python3
"""tools/exp/step_timeline.py <kernel_trace.csv> <n_launches_per_step>: the last step's kernels in launch order with their
durations and the gap to the previous kernel (us) -- where a launch-bound step spends its time."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
rows = rows[-n:]
prev_end = None
tot = gap_tot = 0.0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
    name = r["Kernel_Name"]
    name = name[:70]
    print(f"{(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  {name}")
    tot += (e - s) / 1e3
    gap_tot += max(gap, 0.0)
    prev_end = e
print(f"kernels {tot:.1f} us, gaps {gap_tot:.1f} us, span {(int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3:.1f} us")
