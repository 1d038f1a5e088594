#!/bin/bash
# Verdict r5 item 4: the WEIGHT operand only as pre-split bf16 planes (written once per step by Adam), staged global -> LDS by DMA, the
# activation operand on the shipped path (fp32 loads -> split under the MFMAs -> LDS).  Timing-only ablation, results wrong by design.
#   git apply tools/exp/dma_bplanes_ablation.patch            (NEVER commit the patched kernel)
#   tools/exp/build_variant.sh ablb -DHYPEL_ABL=2048;  git checkout hypelcnn_amd/csrc/seg_gemm.hip
#   tools/gpu.sh --timeout 600 -- 'bash tools/exp/dma_bplanes_ablation.sh'
# Form (layout nt = the data gradients, 128 x 128 block / 512 threads): LDS = two A buffers of 18 KB (as shipped) + THREE B stages of
# 12 KB (3 planes x 128 rows x 32 bytes, unpadded, swizzled fragment reads) = the shipped 72 KB, two blocks per CU as shipped (96 vs 105
# VGPRs).  Per k-tile and thread: one 16-byte A load (as shipped) + 1.5 16-byte `buffer_load ... lds` transfers (waves 0-3 two, waves
# 4-7 one: 12 KB = 1.5 x the fp32 bytes of the B tile), no B staging registers, no B split arithmetic (half of the block's), no B
# ds_write.  Every vector memory operation of the loop is issued from inline assembly and ONE hand-placed s_waitcnt per k-tile
# (vmcnt(4) / vmcnt(3)) lets the planes of tile t + 2 and the A loads of tiles t + 3, t + 4 fly across the barrier (ISA checked) --
# the prefetch depth of the shipped A ring is kept, so the form is not handicapped the way r5's compiler-issued one was.
# In its favour and NOT charged: the planes are read from the fp32 weight rows as 64 contiguous bytes (real planes: 32-byte pieces per
# row and plane unless stored k-tile-major); Adam's 2 x 6 extra bytes per parameter (both orientations); the pack kernels of the merged
# levels would have to emit planes too.
# THRESHOLD, written before the run (verdict): build only if K = 480 shows <= -6 % (median of the interleaved passes).  Reading for
# the headline if it does: the three merged data gradients + the forward launches that would take the form are ~2.2 ms of the step.
for pass in 1 2 3; do for lib in default ablb; do
  if [ $lib = default ]; then unset HYPEL_LIB_PATH; else export HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/variants/$lib/libhypel_hip.so; fi
  for k in 480 1920; do
    echo "$lib pass $pass $(timeout 90 python tools/exp/split_probe.py --m 50176 --k $k --n 480 --layout nt --variants 3 --reps 7 2>&1 | grep layout | sed 's/layout nt M=50176 //')"
  done
done; done
# the headline step with the ablation library (its data gradients of the nt 128 x 128 form only; numbers wrong, timing only)
for pass in 1 2; do for lib in default ablb; do
  if [ $lib = default ]; then unset HYPEL_LIB_PATH; else export HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/variants/$lib/libhypel_hip.so; fi
  echo "step $lib pass $pass $(timeout 200 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-input-pipeline 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["gemm_ms_per_step"] if "gemm_ms_per_step" in d["roofline"] else "")' 2>&1 | tail -1)"
done; done
