#!/usr/bin/env python3
"""Compact view of a rocprofv3 kernel_stats.csv: calls, average us, share -- kernel names shortened."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in rows[:top]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    print(f"{name[:70]:70s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:9.2f} ms  {float(r['Percentage']):5.2f} %")
