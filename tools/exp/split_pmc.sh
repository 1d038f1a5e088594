#!/bin/bash
# SQ counters of the split kernel vs the fp32 kernel on one product (tools/exp/split_probe.py).
#   tools/gpu.sh --timeout 900 -- 'bash tools/exp/split_pmc.sh "--m 50176 --k 480 --n 480 --layout nn --variants 0,3"'
ARGS=${1:-"--m 50176 --k 480 --n 480 --layout nn --variants 0,3"}
mkdir -p gpurun_out/split_pmc
python tools/exp/split_probe.py $ARGS --reps 7 | tee gpurun_out/split_pmc/timing.txt
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc_sp
  rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_sp -o q -- python $GRAFT_REPO_ROOT/tools/exp/split_probe.py $ARGS --reps 2 > /dev/null 2> /tmp/sp.err
  Q=$(find /tmp/pmc_sp -name "*counter_collection.csv" | head -1)
  python - "$Q" <<'PY' | tee -a $GRAFT_REPO_ROOT/gpurun_out/split_pmc/counters.txt
import csv, sys, collections
disp = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "seg_gemm" not in r["Kernel_Name"]: continue
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"][:90], "vgpr": r.get("VGPR_Count"), "lds": r.get("LDS_Block_Size"), "grid": r.get("Grid_Size"), "wg": r.get("Workgroup_Size")})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
last = {}
for k, d in disp.items():
    last[d["name"]] = d
for name, d in last.items():
    print(name, "vgpr", d["vgpr"], "lds", d["lds"], "grid", d["grid"], "wg", d["wg"])
    for c, v in d.items():
        if c not in ("name", "vgpr", "lds", "grid", "wg"): print(f"    {c:28s} {v:16.0f}")
PY
done
