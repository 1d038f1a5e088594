#!/bin/bash
# register-weight forward of the narrow dense stack (DS_REGW) vs the LDS-staged kernel: parity first, then launch times and the CycleGAN step
#   FILE=dense_stack.hip tools/exp/build_variant.sh noregw -DDS_REGW=0
#   tools/gpu.sh --timeout 1200 -- 'bash tools/exp/dense_regw_ab.sh'
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gan.py -x -q -m gpu -k "dense_stack or cycle or gan_x2y or dcl or feature" 2>&1 | tail -2
for pass in 1 2; do for lib in default noregw; do
  if [ $lib = default ]; then unset HYPEL_LIB_PATH; else export HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/variants/$lib/libhypel_hip.so; fi
  echo "== $lib pass $pass"
  python tools/exp/dense_stack_probe.py 2>/dev/null | grep widths
  python bench.py --workload cyclegan --steps 2000 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cyclegan', round(d['ms_per_step'],5), 'ms')"
done; done
