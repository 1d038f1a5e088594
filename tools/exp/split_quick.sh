#!/bin/bash
# quick loop: split-kernel parity tests + the probe on the three layouts + the step's per-launch A/B
mkdir -p gpurun_out/split_ab
o=gpurun_out/split_ab
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "split6" > $o/tests.txt 2>&1
tail -3 $o/tests.txt
python tools/exp/split_probe.py --m 50176 --k 480 --n 480 --layout nn --variants ${PV:-0,2,3} 2>&1 | grep layout
python tools/exp/split_probe.py --m 50176 --k 480 --n 480 --layout nt --variants ${PV:-0,2,3} 2>&1 | grep layout
python tools/exp/split_probe.py --m 480 --k 50176 --n 480 --layout tn --variants ${PV:-0,2,3} 2>&1 | grep layout
python tools/exp/split_probe.py --m 50176 --k 240 --n 240 --layout nn --variants ${PV:-0,2,3} 2>&1 | grep layout
for v in ${VARIANTS:-0 6}; do
  HYPEL_GEMM_SPLIT=$v HYPEL_GEMM_SPLIT_MIN_GFLOP=${MINGF:-2} python tools/gemm_microbench.py --rounds ${ROUNDS:-12} > $o/mb_$v.txt 2>&1
  tail -1 $o/mb_$v.txt
done
