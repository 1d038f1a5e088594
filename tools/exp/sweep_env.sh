#!/bin/bash
# tools/exp/sweep_env.sh VAR v1 v2 ...  [-- extra env assignments]: one bench line (ms/step, GEMM ms/step) per value, two rounds
VAR=$1; shift
BENCH="python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-input-pipeline"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), round(d["roofline"]["gemm_ms_per_step"],4))'
for r in 1 2; do for v in "$@"; do env $VAR=$v $BENCH 2>/dev/null | python -c "$P" "$VAR=$v"; done; done
