#!/usr/bin/env python3
"""tools/exp/last_launches.py <kernel_trace.csv> <name substring> <n>: durations (us) of the last n launches whose kernel
name contains the substring, in launch order (one step's worth of a kernel, launch by launch)."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[3])
print(" ".join(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f}" for r in rows[-n:]))
