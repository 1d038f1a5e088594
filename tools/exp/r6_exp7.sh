#!/bin/bash
# round 6, experiment 7: new tests (hard operands, non-finite, two-rank SyncBN on the GPU), taps per tile of the unmerged 60-filter level
o=gpurun_out/r6_exp7; mkdir -p $o
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "hard_operands or nonfinite or kslice" > $o/tests_k.txt 2>&1; tail -2 $o/tests_k.txt
python -m pytest tests/test_gpu_dp.py -x -q -m gpu > $o/tests_dp.txt 2>&1; tail -3 $o/tests_dp.txt
python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "dualcnn_large_batches" > $o/tests_dual.txt 2>&1; tail -3 $o/tests_dual.txt
python -m pytest tests/test_gpu_gan.py -x -q -m gpu > $o/tests_gan.txt 2>&1; tail -2 $o/tests_gan.txt
MB="python tools/gemm_microbench.py --rounds 8 --with-reduce --filter fwd:connector_"
i=0
for ps in "MERGE_MAX_TAPS=0" "MAX_TAPS_PER_TILE=16" "MAX_TAPS_PER_TILE=25" "MAX_TAPS_PER_TILE=13"; do
  HYPEL_PLAN_SET=$ps $MB > $o/t$i.txt 2>&1
  echo "== t$i $ps"; grep -h "fwd:connector_[012]_\|tap-split" $o/t$i.txt | cut -c1-118
  i=$((i+1))
done
for w in cut cyclegan; do python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])' $w; done
HYPEL_PLAN_SET=SPLIT_NOMINAL_BATCH_GAN=1024 python bench.py --workload cut --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("cut nominal 1024", d["ms_per_step"], d["value"])'
