#!/bin/bash
# NOTES 4.C: forward 1x1 GEMMs with the consumer-side BN + activation (+ shortcut, + activation store) probe builds
# (tools/exp/mkalt.sh probeN "-DHYPEL_ASTAGE_PROBE=N"); timing only, the probe's results are garbage
for n in tree probe1 probe2 probe3; do
  echo "== $n"
  if [ $n = tree ]; then L=""; else L="HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/alt/libhypel_$n.so"; fi
  env $L python tools/gemm_microbench.py --rounds 12 --filter fwd:con 2>/dev/null | grep -E "conv_enc|conv_dec|connector_conv|TOTAL"
done
