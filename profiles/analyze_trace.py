"""Per-launch table of one training step from a rocprofv3 kernel trace of bench.py:
  python profiles/analyze_trace.py <kernel_trace.csv> [all] [--workload hypelcnn|dualcnn]
The launches of the LAST full step in the trace are matched, in order, with the planner's launch list (built here on
the CPU emulation backend), so every kernel gets its layer tag and algorithmic FLOP."""
import csv, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
workload = "hypelcnn"
if "--workload" in sys.argv:
    i = sys.argv.index("--workload")
    workload = sys.argv[i + 1]
    del sys.argv[i:i + 2]
trace = sys.argv[1]
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# plan
from tests.emu_backend import EmuBackend
import bench
nb = bench.CLASSIFIER_WORKLOADS[workload][5]
ctx, train_step, lr, alg = bench.build_model(nb, EmuBackend(), workload)
ctx.capture_graphs = False
ct = train_step.compiled(nb)
plan = ct.plan
launches = [l for l in plan.fwd + plan.bwd if l.name not in ('_fork','_join')]
names = []
for l in launches:
    names.append(l.name)
    if l.name in ("mse", "sum_f32"): names.append(l.name + "_fin")
# find the last full step in the trace: locate sequences starting with nhwc_to_pnc
kn = [r['Kernel_Name'] for r in rows]
starts = [i for i, k in enumerate(kn) if 'nhwc_to_pnc' in k]
i0 = starts[-1]
seq = rows[i0:]
# walk: match only hypel kernels (skip torch/copy)
def is_h(k): return 'anonymous namespace' in k
seq = [r for r in seq if is_h(r['Kernel_Name'])]
out = []
j = 0
for l in launches:
    n_k = 2 if l.name in ("mse", "sum_f32") else 1
    dur = 0
    for _ in range(n_k):
        r = seq[j]; j += 1
        dur += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    out.append((l.name, l.tag, l.flops, dur, r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size')))
tot = sum(o[3] for o in out)
print("total kernel ns per step", tot, " gemm ns", sum(o[3] for o in out if o[0].startswith('seg_gemm')))
print(f"{'name':18s} {'tag':34s} {'GFLOP':>8s} {'us':>8s} {'TF/s':>7s} grid")
for o in out:
    if o[3] > 20000 or (len(sys.argv) > 2 and sys.argv[2] == 'all'):
        tf = o[2]/o[3]/1e3 if o[2] else 0
        print(f"{o[0]:18s} {o[1]:34s} {o[2]/1e9:8.2f} {o[3]/1e3:8.1f} {tf:7.1f} {o[4]}")
bytag = {}
for o in out:
    key = o[0] if not o[0].startswith('seg_gemm') else 'gemm:' + o[1].split(':')[0]
    bytag.setdefault(key, [0, 0, 0]); bytag[key][0] += o[3]; bytag[key][1] += o[2]; bytag[key][2] += 1
for k, v in sorted(bytag.items(), key=lambda t: -t[1][0]):
    print(f"{k:28s} {v[0]/1e3:9.1f} us  n={v[2]:3d}  {v[1]/max(v[0],1)/1e3:6.1f} TF/s")
