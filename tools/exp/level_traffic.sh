#!/bin/bash
# tools/exp/level_traffic.sh VAR v1 v2 ...: per-launch fabric traffic of the multi-kernel levels' forward launches (PMC passes
# FETCH_SIZE / WRITE_SIZE of an eager headline step) under an environment setting, next to the step time
export TMPDIR=/tmp
ROOT=$(pwd)
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  rm -rf /tmp/lt_f /tmp/lt_w
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/lt_f -o f -- python $ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-input-pipeline > /dev/null 2>&1)
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/lt_w -o w -- python $ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-input-pipeline > /dev/null 2>&1)
  F=$(find /tmp/lt_f -name "*counter_collection.csv" | head -1); W=$(find /tmp/lt_w -name "*counter_collection.csv" | head -1)
  echo "== $VAR=$v"
  python tools/pmc_traffic_per_launch.py $F $W --workload hypelcnn | grep -E "connector_._conv1x1|^all"
  python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-input-pipeline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step ms', round(d['ms_per_step'],4), 'gemm ms', round(d['roofline']['gemm_ms_per_step'],4))"
done
