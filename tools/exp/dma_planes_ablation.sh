#!/bin/bash
# Operands as pre-split bf16 planes staged global -> LDS by DMA, inside the split loop: timing-only ablation (results wrong by design).
#   git apply tools/exp/dma_planes_ablation.patch               (NEVER commit the patched kernel)
#   tools/exp/build_variant.sh abl512 -DHYPEL_ABL=512; tools/exp/build_variant.sh abl1536 -DHYPEL_ABL=1536;  git checkout hypelcnn_amd/csrc/seg_gemm.hip
#   tools/gpu.sh --timeout 600 -- 'bash tools/exp/dma_planes_ablation.sh'
# Layout nt, 128x128 / 512 threads: three LDS stages of 24 KB (unpadded 32-byte plane rows, swizzled fragment reads), three 16-byte
# buffer_load ... lds per thread and k-tile (1.5 x the fp32 operand bytes), no staging registers / split arithmetic / ds_write.
# KNOWN HANDICAP: hipcc turns the vmcnt(3) in front of the phase barrier into vmcnt(0) -- the transfers of tile t + 2 must land inside
# the phase that issued them (build abl512).  Build abl1536 (HYPEL_ABL & 1024) issues the transfers from inline assembly, which the
# s_waitcnt insertion does not see: vmcnt(3) stays, tile t + 2 may fly across the barrier (ISA checked; tools/exp/probe/lds_dma16_asm_probe.hip).
for pass in 1 2; do for lib in default abl512 abl1536; do
  if [ $lib = default ]; then unset HYPEL_LIB_PATH; else export HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/variants/$lib/libhypel_hip.so; fi
  for k in 480 1920; do
    echo "$lib pass $pass $(timeout 90 python tools/exp/split_probe.py --m 50176 --k $k --n 480 --layout nt --variants 3 --reps 7 2>&1 | grep layout | sed 's/layout nt M=50176 //')"
  done
done; done
