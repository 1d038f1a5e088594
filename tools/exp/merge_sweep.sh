#!/bin/bash
# per-launch A/B of the merged-level variants (tools/gemm_microbench.py, level launches only)
run() { echo "== $*"; env "$@" python tools/gemm_microbench.py --rounds 10 2>/dev/null | grep -E "connector_[0-9]_conv|wgrad|TOTAL"; }
run HYPEL_MERGE_LEVELS=0
run HYPEL_MERGE_LEVELS=fwd HYPEL_MERGE_FWD_MAX_COUT=16 HYPEL_MERGE_MAX_TAPS=12
run HYPEL_MERGE_LEVELS=fwd HYPEL_MERGE_FWD_MAX_COUT=16 HYPEL_MERGE_MAX_TAPS=16
run HYPEL_MERGE_LEVELS=fwd HYPEL_MERGE_FWD_MAX_COUT=16 HYPEL_MERGE_MAX_TAPS=24
run HYPEL_MERGE_LEVELS=fwd HYPEL_MERGE_FWD_HINT=1
run HYPEL_MERGE_LEVELS=wgrad HYPEL_MERGE_WGRAD_MAX_COUT=16
run HYPEL_MERGE_LEVELS=fwd HYPEL_MERGE_LEVELS_MAX_COUT=64 HYPEL_MERGE_FWD_HINT=3
