#!/bin/bash
# Would producer-side bf16 planes pay under the power cap?  First-order, timing-only (results wrong by design; scratch builds with
# -DHYPEL_ABL: 1 = no split arithmetic (planes = raw bits), 256 = +50 % operand bytes per 16-byte load, 257 = both = what a kernel
# fed with pre-split planes through registers would move and compute).  Layout nt: both operands k-contiguous.
#   git apply tools/exp/planes_energy_ablation.patch            (timing-only code: NEVER commit the patched kernel)
#   for a in 1 256 257; do tools/exp/build_variant.sh abl$a -DHYPEL_ABL=$a; done;  git checkout hypelcnn_amd/csrc/seg_gemm.hip
#   tools/gpu.sh --timeout 900 -- 'bash tools/exp/planes_energy_ablation.sh'
for pass in 1 2; do for lib in default abl1 abl256 abl257; do
  if [ $lib = default ]; then unset HYPEL_LIB_PATH; else export HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/variants/$lib/libhypel_hip.so; fi
  for k in 480 1920; do
    echo "$lib pass $pass $(python tools/exp/split_probe.py --m 50176 --k $k --n 480 --layout nt --variants 3 --reps 7 2>&1 | grep layout | sed 's/layout nt M=50176 //')"
  done
done; done
