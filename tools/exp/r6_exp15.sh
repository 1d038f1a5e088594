#!/bin/bash
# NEEDS tools/exp/bnsum_epilogue.patch applied (git apply; the form was measured neutral and is not in the tree) and the library rebuilt
# the two forms of the BSUM epilogue (y tile requested before / after the tile's stores) against the plan without it, one box
o=gpurun_out/r6_exp15; mkdir -p $o
for pass in 1 2 3; do for cfg in "off" "early" "late"; do
  unset HYPEL_LIB_PATH; set=BNSUM_EPILOGUE=1
  [ $cfg = off ] && set=BNSUM_EPILOGUE=0
  [ $cfg = late ] && export HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/variants/bslate/libhypel_hip.so
  echo "$cfg pass $pass $(HYPEL_PLAN_SET=$set timeout 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-input-pipeline 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"].get("gemm_ms_per_step"))' 2>&1 | tail -1)"
done; done | tee $o/step_ab.txt
