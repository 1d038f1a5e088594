#!/bin/bash
# tools/exp/ab_libs.sh name1 name2 ...   (alt/libhypel_<name>.so; "tree" = the in-tree build), one bench line each, 2 rounds
BENCH="python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-input-pipeline"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), round(d["ms_per_step_median"],4), round(d["roofline"]["gemm_ms_per_step"],4))'
for r in 1 2; do for n in "$@"; do
  if [ "$n" = tree ]; then $BENCH 2>/dev/null | python -c "$P" tree
  else HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/alt/libhypel_$n.so $BENCH 2>/dev/null | python -c "$P" $n; fi
done; done
