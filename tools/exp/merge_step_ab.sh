#!/bin/bash
# step-level A/B of the merged-level passes (same box, interleaved)
run() { printf "%-90s " "$*"; env "$@" python bench.py --steps 60 --no-cpu-baseline --no-input-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('%.3f ms  frac %.4f  gemm %.3f ms  %d launches' % (d['ms_per_step'], r['frac'], r['gemm_ms_per_step'], r['kernel_launches_per_step']))"; }
for rep in 1 2; do
run HYPEL_MERGE_LEVELS=0
run HYPEL_MERGE_LEVELS=fwd HYPEL_MERGE_FWD_MAX_COUT=16
run HYPEL_MERGE_LEVELS=dgrad
run HYPEL_MERGE_LEVELS=fwd,dgrad HYPEL_MERGE_FWD_MAX_COUT=16
run HYPEL_MERGE_LEVELS=fwd,dgrad HYPEL_MERGE_FWD_MAX_COUT=16 HYPEL_MERGE_DGRAD_MAX_COUT=16
run HYPEL_MERGE_LEVELS=fwd,dgrad,wgrad HYPEL_MERGE_FWD_MAX_COUT=16 HYPEL_MERGE_WGRAD_MAX_COUT=16
done
