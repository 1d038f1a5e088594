#!/bin/bash
# the shader clock the split kernel's blocks run at (library built with -DHYPEL_GEMM_CLK=1), per build in $1
for lib in ${1:-clk}; do
  export HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/variants/$lib/libhypel_hip.so
  for v in 0 3; do
  echo "== $lib variant $v"
  python tools/exp/split_probe.py --m 50176 --k 480 --n 480 --layout nn --variants $v --clk 2>&1 | grep -v amdgpu.ids
  python tools/exp/split_probe.py --m 50176 --k 1920 --n 480 --layout nn --variants $v --clk 2>&1 | grep -v amdgpu.ids
  done
done
