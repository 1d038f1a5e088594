#!/bin/bash
# NEEDS tools/exp/wide64_on_128_blocks.patch (experiment constant EXP_WIDE64 + variant library wide64)
# 33 .. 64-column split products (60-filter level forward, the s64 class of the merged filter gradients) on 128 x 128 blocks with half of
# the waves idle (64 x 32 wave tiles: 9 fragment reads per 12 MFMAs) instead of 128 x 64 blocks (32 x 32 wave tiles: 6 per 6)
for pass in 1 2; do for c in 0 1; do
  unset HYPEL_LIB_PATH; [ $c = 1 ] && export HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/variants/wide64/libhypel_hip.so
  echo "== EXP_WIDE64=$c pass $pass"
  HYPEL_PLAN_SET=EXP_WIDE64=$c timeout 300 python tools/gemm_microbench.py --filter "connector_0_conv1x1" 2>&1 | grep "GF" | cut -c1-130
  HYPEL_PLAN_SET=EXP_WIDE64=$c timeout 300 python tools/gemm_microbench.py --filter "wgrad" 2>&1 | grep "GF" | cut -c1-130
  echo "step EXP_WIDE64=$c pass $pass $(HYPEL_PLAN_SET=EXP_WIDE64=$c timeout 200 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-input-pipeline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"].get("gemm_ms_per_step"))')"
done; done
