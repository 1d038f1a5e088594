#!/bin/bash
# per-launch tile width of the split kernels where the drain probe (tools/exp/drain_probe.py) shows a launch that does not fill the
# 512 resident slots or ends in a long drain: 128-wide (default for n > 96) vs 64-wide, same box, alternating
#   tools/gpu.sh --timeout 1500 -- 'bash tools/exp/width_override_ab.sh'
A="dgrad:connector_0_conv1x1=2"
B="$A,dgrad:connector_1_conv1x1/merged=2"
C="$B,dgrad:conv_enc_2=2,fwd:conv_dec_1=2,fwd:connector_conv_0=2,dgrad:connector_conv_0=2,fwd:conv_enc_1=2,dgrad:conv_dec_2=2"
for pass in 1 2; do for name in base A B C; do
  case $name in base) export HYPEL_SPLIT_OVERRIDE="";; A) export HYPEL_SPLIT_OVERRIDE="$A";; B) export HYPEL_SPLIT_OVERRIDE="$B";; C) export HYPEL_SPLIT_OVERRIDE="$C";; esac
  echo "$name pass $pass: $(python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-input-pipeline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'ms')")"
done; done
for name in base C; do
  case $name in base) export HYPEL_SPLIT_OVERRIDE="";; C) export HYPEL_SPLIT_OVERRIDE="$C";; esac
  echo "== per launch, $name"
  python tools/gemm_microbench.py --rounds 12 2>/dev/null | grep -E "dgrad:connector_0_conv1x1|dgrad:connector_1_conv1x1|dgrad:conv_enc_2|fwd:conv_dec_1|connector_conv_0|fwd:conv_enc_1 |dgrad:conv_dec_2|TOTAL" | cut -c1-120
done
