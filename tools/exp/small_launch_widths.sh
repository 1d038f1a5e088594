#!/bin/bash
# per-launch A/B of tile widths for the three sub-65 TF/s launches (HYPEL_HINT_OVERRIDE; 1 = 128x32, 2 = 128x64, 3 = 128x96)
for h in 1 2 3; do
  echo "== width hint $h"
  HYPEL_HINT_OVERRIDE="fwd:conv_enc_0=$h,fwd:connector_conv_1=$h,dgrad:connector_conv_1=$h,fwd:conv_enc_1=$h,fwd:conv_dec_2=$h,dgrad:conv_dec_2=$h,dgrad:conv_enc_1=$h" \
    python tools/gemm_microbench.py --rounds 12 2>/dev/null | grep -E "conv_enc_0|connector_conv_1|conv_enc_1|conv_dec_2"
done
