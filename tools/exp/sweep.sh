#!/bin/bash
# tools/exp/sweep.sh "ENV1=a ENV2=b" "ENV1=c" ...   one bench line per setting (same box)
BENCH="python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-input-pipeline"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "|", round(d["ms_per_step"],4), round(d["ms_per_step_median"],4), d["roofline"]["launches_per_step"], round(d["roofline"]["gemm_ms_per_step"],4))'
for s in "$@"; do
  env $s $BENCH 2>/dev/null | python -c "$P" "$s"
done
