#!/bin/bash
# Round 5: the split-operand GEMM kernels (HYPEL_GEMM_SPLIT6) -- kernel parity tests, then a per-launch A/B of the benchmark
# step's GEMM launches: fp32 MFMA kernels vs the split kernels at each block width.
#   tools/gpu.sh --timeout 1500 -- 'bash tools/exp/split_ab.sh'
mkdir -p gpurun_out/split_ab
o=gpurun_out/split_ab
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "split6" -s > $o/tests.txt 2>&1
tail -5 $o/tests.txt
for v in ${VARIANTS:-0 6:2 6:3 6}; do
  HYPEL_GEMM_SPLIT=$v HYPEL_GEMM_SPLIT_MIN_GFLOP=${MINGF:-0} python tools/gemm_microbench.py --rounds ${ROUNDS:-12} > $o/mb_$v.txt 2>&1
  tail -1 $o/mb_$v.txt
done
