#!/bin/bash
# round 6, experiment 3: merged levels on the split kernels -- taps per tile without channel parts; level 0 (60 filters) merged too
o=gpurun_out/r6_exp3; mkdir -p $o
MB="python tools/gemm_microbench.py --rounds 8 --with-reduce --filter connector_"
i=0
for ps in "MERGE_MAX_TAPS=0" \
          "MERGE_FWD_MAX_COUT_SPLIT=32,MERGE_MAX_TAPS=16,L2_CHUNK_BYTES=33554432" \
          "MERGE_FWD_MAX_COUT_SPLIT=32,MERGE_MAX_TAPS=25,L2_CHUNK_BYTES=33554432" \
          "MERGE_LEVELS_MAX_COUT=64,MERGE_FWD_MAX_COUT_SPLIT=64,MERGE_MAX_TAPS=16,L2_CHUNK_BYTES=33554432" \
          "MERGE_LEVELS_MAX_COUT=64,MERGE_FWD_MAX_COUT_SPLIT=64,MERGE_MAX_TAPS=25,L2_CHUNK_BYTES=33554432" \
          "MERGE_LEVELS_MAX_COUT=64,MERGE_FWD_MAX_COUT_SPLIT=32,MERGE_MAX_TAPS=16,L2_CHUNK_BYTES=33554432"; do
  HYPEL_PLAN_SET=$ps $MB > $o/c$i.txt 2>&1
  echo "== c$i $ps"; grep -h "connector_[012]_\|tap-split\|dgrad-split" $o/c$i.txt | cut -c1-118
  i=$((i+1))
done
