#!/bin/bash
run() { printf "%-60s " "$*"; env "$@" python bench.py --steps 60 --no-cpu-baseline --no-input-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('%.3f ms  median %.3f' % (d['ms_per_step'], d['ms_per_step_median']))"; }
run X=0
run HYPEL_STAT_BLOCKS=128
run HYPEL_STAT_BLOCKS=512
run HYPEL_STAT_BLOCKS=1024 HYPEL_STAT_CHUNK_ROWS=256
run HYPEL_STAT_CHUNK_ROWS=512 HYPEL_STAT_BLOCKS=128
run HYPEL_RED_GX=512
run HYPEL_RED_GX=1024
run X=0
