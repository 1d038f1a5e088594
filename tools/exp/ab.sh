#!/bin/bash
# same-box A/B of an environment switch:  tools/exp/ab.sh VAR A_VALUE B_VALUE [rounds]
VAR=$1; A=$2; B=$3; R=${4:-2}
BENCH="python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-input-pipeline"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), round(d["ms_per_step_median"],4), d["roofline"]["launches_per_step"], round(d["roofline"]["gemm_ms_per_step"],4))'
for i in $(seq $R); do
  env $VAR=$A $BENCH 2>/dev/null | python -c "$P" "$VAR=$A"
  env $VAR=$B $BENCH 2>/dev/null | python -c "$P" "$VAR=$B"
done
