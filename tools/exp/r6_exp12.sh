#!/bin/bash
# round 6, experiment 12: early side-stream flushes of the merged filter gradients (WGRAD_OVERLAP)
o=gpurun_out/r6_exp12; mkdir -p $o
HYPEL_PLAN_SET=WGRAD_OVERLAP=1 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "batch1024_vs_oracle or hip_graph_replay or batch1024_properties or end_to_end" > $o/tests_m.txt 2>&1; tail -3 $o/tests_m.txt
BENCH="python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-input-pipeline"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), round(d["ms_per_step_median"],4), round(d["roofline"]["gemm_ms_per_step"],4))'
for r in 1 2; do
  $BENCH 2>/dev/null | python -c "$P" off
  HYPEL_PLAN_SET=WGRAD_OVERLAP=1 $BENCH 2>/dev/null | python -c "$P" overlap-0.45
  HYPEL_PLAN_SET=WGRAD_OVERLAP=1,WGRAD_OVERLAP_FRAC_A=0.3 $BENCH 2>/dev/null | python -c "$P" overlap-0.3
  HYPEL_PLAN_SET=WGRAD_OVERLAP=1,WGRAD_OVERLAP_FRAC_A=0.6 $BENCH 2>/dev/null | python -c "$P" overlap-0.6
  HYPEL_PLAN_SET=WGRAD_OVERLAP=1,WGRAD_OVERLAP_FRAC_A=0.3,WGRAD_OVERLAP_FRAC_B=0.65 $BENCH 2>/dev/null | python -c "$P" overlap-0.3+0.65
  HYPEL_PLAN_SET=WGRAD_OVERLAP=1,WGRAD_OVERLAP_FRAC_A=0.15,WGRAD_OVERLAP_FRAC_B=0.5 $BENCH 2>/dev/null | python -c "$P" overlap-0.15+0.5
done
