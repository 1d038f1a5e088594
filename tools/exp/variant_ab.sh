#!/bin/bash
# same-box A/B of library builds: kernel parity tests on the default build, then the probe products and the step's
# per-launch microbench on every build named in LIBS (default = the in-tree library, NAME = csrc/variants/NAME)
#   tools/gpu.sh --timeout 1500 -- 'LIBS="default nodot" bash tools/exp/variant_ab.sh'
mkdir -p gpurun_out/variant_ab
o=gpurun_out/variant_ab
for lib in ${LIBS:-default}; do
  if [ $lib = default ]; then unset HYPEL_LIB_PATH; else export HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/variants/$lib/libhypel_hip.so; fi
  python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "split6" > $o/tests_$lib.txt 2>&1
  echo "tests $lib: $(tail -1 $o/tests_$lib.txt)"
done
for r in 1 2; do
for lib in ${LIBS:-default}; do
  if [ $lib = default ]; then unset HYPEL_LIB_PATH; else export HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/variants/$lib/libhypel_hip.so; fi
  echo "== $lib (pass $r)"
  python tools/exp/split_probe.py --m 50176 --k 480 --n 480 --layout nn --variants 3 2>&1 | grep layout
  python tools/exp/split_probe.py --m 50176 --k 480 --n 480 --layout nt --variants 3 2>&1 | grep layout
  python tools/exp/split_probe.py --m 480 --k 50176 --n 480 --layout tn --variants 3 2>&1 | grep layout
  python tools/gemm_microbench.py --rounds ${ROUNDS:-12} > $o/mb_${lib}_$r.txt 2>&1
  tail -1 $o/mb_${lib}_$r.txt
  if [ -n "$DUAL" ]; then python bench.py --workload dualcnn --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200; fi
done
done
