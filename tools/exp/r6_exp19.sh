#!/bin/bash
# the driver's own command line (BENCH_r05.json: --gpus 1 --steps 20 --warmup 5) next to the default, same box
for i in 1 2 3; do
  echo "driver-form $i $(python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], d["ms_per_step_median"], d["value"], r["sustained_clock_ghz"], r["sustained_power_w"], d["cpu_baseline"]["value"])')"
  echo "default     $i $(python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["ms_per_step_median"], d["value"])')"
done
