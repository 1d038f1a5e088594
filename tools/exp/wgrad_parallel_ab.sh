#!/bin/bash
run() { printf "%-70s " "$*"; env "$@" python bench.py --steps 60 --no-cpu-baseline --no-input-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('%.3f ms  median %.3f  gemm %.3f ms' % (d['ms_per_step'], d['ms_per_step_median'], r['gemm_ms_per_step']))"; }
for rep in 1 2; do
run HYPEL_WGRAD_PARALLEL=0
run HYPEL_WGRAD_PARALLEL=1
run HYPEL_MERGE_LEVELS_MAX_COUT=64
run HYPEL_WGRAD_PARALLEL=1 HYPEL_MERGE_LEVELS_MAX_COUT=64
done
