#!/bin/bash
# round 6, experiment 8: the 4 x 2 / 32x64 split variant for merged forward launches (VAR_N): tests, then A/B against the 2 x 4 / 64x32 deal
o=gpurun_out/r6_exp8; mkdir -p $o
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "hard_operands or nonfinite or per_tile_column or split6" > $o/tests_k.txt 2>&1; tail -2 $o/tests_k.txt
python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "batch1024_vs_oracle or every_level" > $o/tests_m.txt 2>&1; tail -2 $o/tests_m.txt
MB="python tools/gemm_microbench.py --rounds 10 --with-reduce --filter fwd:connector_"
for r in 1 2; do
for lib in default novarn; do
  if [ $lib = default ]; then unset HYPEL_LIB_PATH; else export HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/variants/$lib/libhypel_hip.so; fi
  $MB > $o/mb_${lib}_$r.txt 2>&1; echo "== $lib $r: $(grep -h 'fwd:connector_1' $o/mb_${lib}_$r.txt | cut -c1-110)"
  HYPEL_PLAN_SET=MERGE_LEVELS_MAX_COUT=64,MERGE_FWD_MAX_COUT_SPLIT=64 $MB > $o/mb64_${lib}_$r.txt 2>&1; echo "== $lib $r level0 merged: $(grep -h 'fwd:connector_0\|tap-split-reduce#[123] ' $o/mb64_${lib}_$r.txt | cut -c1-100 | tr '\n' '|')"
done
done
unset HYPEL_LIB_PATH
BENCH="python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-input-pipeline"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), round(d["ms_per_step_median"],4), round(d["roofline"]["gemm_ms_per_step"],4))'
for r in 1 2; do
  $BENCH 2>/dev/null | python -c "$P" varn-kernel
  HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/variants/novarn/libhypel_hip.so $BENCH 2>/dev/null | python -c "$P" novarn
done
