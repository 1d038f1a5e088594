import csv, sys, collections
sys.path.insert(0, "/root/repo")
rows = list(csv.DictReader(open(sys.argv[1])))
# group by dispatch
disp = collections.OrderedDict()
for r in rows:
    d = disp.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'], 'start': int(r['Start_Timestamp']), 'end': int(r['End_Timestamp']), 'vgpr': r['VGPR_Count'], 'agpr': r['Accum_VGPR_Count'], 'grid': r['Grid_Size'], 'lds': r['LDS_Block_Size']})
    d[r['Counter_Name']] = float(r['Counter_Value'])
ds = [d for _, d in sorted(disp.items())]
from tests.emu_backend import EmuBackend
import bench
ctx, ts, lr, alg = bench.build_model(1024, EmuBackend()); ctx.capture_graphs = False
plan = ts.compiled(1024).plan
launches = [l for l in plan.fwd + plan.bwd if l.name not in ('_fork', '_join')]
hk = [d for d in ds if 'anonymous namespace' in d['name']]
starts = [i for i, d in enumerate(hk) if 'nhwc_to_pnc' in d['name']]
seq = hk[starts[-1]:]
j = 0
print(f"{'tag':34s} {'GF':>6s} {'us':>7s} {'TF':>5s} {'cyc/MFMA':>8s} {'mfma_busy%':>10s} {'wait_any%':>9s} {'wait_inst%':>10s} vgpr agpr lds grid")
for l in launches:
    n_k = 2 if l.name in ("mse", "sum_f32") else 1
    d = seq[j]; j += n_k
    if not l.name.startswith('seg_gemm'): continue
    dur = d['end'] - d['start']
    n_mfma = l.flops / 8192
    wc = d['SQ_WAVE_CYCLES']
    # busy% assuming 1024 SIMDs and 2.1 GHz
    busy = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (dur * 1e-9 * 1024 * 2.1e9)
    print(f"{l.tag:34s} {l.flops/1e9:6.2f} {dur/1e3:7.1f} {l.flops/dur/1e3:5.1f} {d['SQ_VALU_MFMA_BUSY_CYCLES']/n_mfma:8.1f} {100*busy:10.1f} {100*d['SQ_WAIT_ANY']/wc:9.1f} {100*d['SQ_WAIT_INST_ANY']/wc:10.1f} {d['vgpr']:>4s} {d['agpr']:>4s} {d['lds']:>5s} {d['grid']}")
