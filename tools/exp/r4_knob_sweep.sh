#!/bin/bash
# re-sweep of the planner knobs whose optimum may have moved with the merged data-gradient segments (same box, one pass)
run() { printf "%-60s " "$*"; env "$@" python bench.py --steps 60 --no-cpu-baseline --no-input-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('%.3f ms  median %.3f  gemm %.3f ms' % (d['ms_per_step'], d['ms_per_step_median'], r['gemm_ms_per_step']))"; }
run X=0
for v in 12 25 49 0; do run HYPEL_DGRAD_MAX_SEGS=$v; done
for v in 7 8 11 13; do run HYPEL_MAX_TAPS=$v; done
for v in 2.5 5; do run HYPEL_L2_CHUNK_MB=$v; done
for v in 1280 2048; do run HYPEL_WGRAD_TARGET_BLOCKS=$v; done
run X=0
