#!/bin/bash
# tools/exp/gen_pmc.sh : SQ counters of the generator kernels in isolation (tools/exp/gen_time.py under rocprofv3 --pmc; two passes)
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rm -rf /tmp/gp_$i
  (cd /tmp && GP_REPS=3 GP_N=${GP_N:-4096,8192} rocprofv3 --pmc $SET --output-format csv -d /tmp/gp_$i -o q -- python $ROOT/tools/exp/gen_time.py > /dev/null 2> /tmp/gp_$i.err)
  tail -2 /tmp/gp_$i.err
  Q=$(find /tmp/gp_$i -name "*counter_collection.csv" | head -1)
  [ -n "$Q" ] && python - "$Q" <<'PY'
import csv, sys, collections
d = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "generator" not in r["Kernel_Name"]: continue
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    k = (int(r["Dispatch_Id"]), nm, r["Grid_Size"])
    d.setdefault(k, {})
    d[k][r["Counter_Name"]] = d[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
# last dispatch of every (kernel, grid)
last = collections.OrderedDict()
for k, v in d.items(): last[(k[1], k[2])] = v
for k, v in last.items():
    print(k[0], "grid", k[1])
    print("   ", "  ".join(f"{n}={x:.4g}" for n, x in v.items()))
PY
done
