#!/bin/bash
# tools/exp/act_in_gemm_ab.sh: CUT / CycleGAN steps with the bias + leaky-ReLU of BN-less dense layers in the product's epilogue
for rep in 1 2; do
for wl in cut cyclegan; do
for h in 1 0; do
  HYPEL_ACT_IN_GEMM=$h python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl act_in_gemm=$h step ms', round(d['ms_per_step'],4), 'launches', d['roofline'].get('launches_per_step'))"
done; done; done
