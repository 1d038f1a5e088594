#!/bin/bash
# same-box A/B of two builds of the library: alt/libhypel_base.so vs the in-tree one
R=${1:-2}
BENCH="python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-input-pipeline"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), round(d["ms_per_step_median"],4), round(d["roofline"]["gemm_ms_per_step"],4))'
for i in $(seq $R); do
  HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/alt/libhypel_base.so $BENCH 2>/dev/null | python -c "$P" base
  $BENCH 2>/dev/null | python -c "$P" new
done
