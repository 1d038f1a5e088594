#!/bin/bash
# round 6, experiment 1: merged forward of the narrow levels on the split kernels, wave deal rows-first vs columns-first
#   tools/gpu.sh --timeout 1200 -- 'bash tools/exp/r6_merge_split_ab.sh'
o=gpurun_out/r6_merge; mkdir -p $o
MB="python tools/gemm_microbench.py --rounds 10 --with-reduce"
for r in 1 2; do
for lib in default colsfirst; do
  if [ $lib = default ]; then unset HYPEL_LIB_PATH; else export HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/variants/$lib/libhypel_hip.so; fi
  $MB > $o/mb_${lib}_base_$r.txt 2>&1; echo "$lib base $r: $(tail -1 $o/mb_${lib}_base_$r.txt)"
  HYPEL_PLAN_SET=MERGE_FWD_MAX_COUT_SPLIT=32,MERGE_FWD_SPLIT_NARROW=1 $MB > $o/mb_${lib}_m32_$r.txt 2>&1; echo "$lib m32 $r: $(tail -1 $o/mb_${lib}_m32_$r.txt)"
done
done
python tools/exp/split_probe.py --m 50176 --k 480 --n 480 --layout nn --variants 3 2>&1 | grep layout
HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/variants/colsfirst/libhypel_hip.so python tools/exp/split_probe.py --m 50176 --k 480 --n 480 --layout nn --variants 3 2>&1 | grep layout
grep -h "connector_[12]\|tap-split" $o/mb_default_base_1.txt $o/mb_default_m32_1.txt $o/mb_colsfirst_m32_1.txt
