#!/bin/bash
# round 6, experiment 5: K-slice records on the data gradients of the split kernels
o=gpurun_out/r6_exp5; mkdir -p $o
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "kslice or split6 or res_epilogue or random_tables or per_tile" > $o/tests_k.txt 2>&1; tail -2 $o/tests_k.txt
python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "batch1024_vs_oracle or every_level or batch64_cfg1 or small_models or hip_graph" > $o/tests_m.txt 2>&1; tail -3 $o/tests_m.txt
MB="python tools/gemm_microbench.py --rounds 8 --with-reduce --filter dgrad"
i=0
for ps in "KSLICE=0" "KSLICE=1" "KSLICE_FRAC_MIN=0.5" "KSLICE_FRAC_MIN=0.7" "KSLICE_MIN_GAIN=0.05" "KSLICE_MIN_GAIN=0.05,KSLICE_FRAC_MIN=0.7"; do
  HYPEL_PLAN_SET=$ps $MB > $o/k$i.txt 2>&1
  echo "== k$i $ps"; grep -h "dgrad\|kslice\|TOTAL" $o/k$i.txt | grep -v "fc_\|image_gen\|conv_2 \|connector_conv_[12]" | cut -c1-118
  i=$((i+1))
done
BENCH="python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-input-pipeline"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), round(d["ms_per_step_median"],4), round(d["roofline"]["gemm_ms_per_step"],4))'
for r in 1 2; do
  HYPEL_PLAN_SET=KSLICE=0 $BENCH 2>/dev/null | python -c "$P" kslice-off
  $BENCH 2>/dev/null | python -c "$P" kslice-default
  HYPEL_PLAN_SET=KSLICE_FRAC_MIN=0.7 $BENCH 2>/dev/null | python -c "$P" kslice-frac0.7
done
