#!/bin/bash
# round 6, experiment 4: merged filter gradients on the split kernels (levels 0 / 1 / 2), then the step
o=gpurun_out/r6_exp4; mkdir -p $o
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "split6 or multi or merged or var_n" > $o/tests_k.txt 2>&1; tail -2 $o/tests_k.txt
python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "batch1024_vs_oracle or every_level or batch64_cfg1 or small_models" > $o/tests_m.txt 2>&1; tail -3 $o/tests_m.txt
MB="python tools/gemm_microbench.py --rounds 8"
i=0
for ps in "MERGE_MAX_TAPS=0" "MERGE_WGRAD_MAX_COUT=0" "MERGE_WGRAD_MAX_COUT=32" "MERGE_WGRAD_MIN_COUT=1"; do
  HYPEL_PLAN_SET=$ps $MB > $o/d$i.txt 2>&1
  echo "== d$i $ps"; grep -h "wgrad\|TOTAL" $o/d$i.txt | cut -c1-118
  i=$((i+1))
done
BENCH="python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-input-pipeline"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), round(d["ms_per_step_median"],4), round(d["roofline"]["gemm_ms_per_step"],4))'
for r in 1 2; do
  $BENCH 2>/dev/null | python -c "$P" new
  HYPEL_PLAN_SET=MERGE_WGRAD_MAX_COUT=0 $BENCH 2>/dev/null | python -c "$P" no-merged-wgrad
  HYPEL_MERGE_LEVELS=fwd,dgrad HYPEL_PLAN_SET=MERGE_LEVELS_MAX_COUT=32,MERGE_FWD_MAX_COUT_SPLIT=16 HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/variants/colsfirst/libhypel_hip.so $BENCH 2>/dev/null | python -c "$P" round5
done
