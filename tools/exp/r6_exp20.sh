#!/bin/bash
# two planner thresholds re-checked on this round's plan, same box, interleaved: the split kernels' size rule and the K-slice bar
run() { echo "$1 $2 pass $3 $(env $1 HYPEL_PLAN_SET=$2 timeout 200 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-input-pipeline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"].get("gemm_ms_per_step"))')"; }
for pass in 1 2; do
  for g in 2 1 1.5 3; do run HYPEL_GEMM_SPLIT_MIN_GFLOP=$g KSLICE_MIN_GAIN=0.10 $pass; done
  for k in 0.05 0.15 0.2; do run HYPEL_GEMM_SPLIT_MIN_GFLOP=2 KSLICE_MIN_GAIN=$k $pass; done
done
