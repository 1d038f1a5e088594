#!/bin/bash
# does the clock / power sampler thread (sysfs reads every 4 ms) cost the launch-bound workloads anything?  same box, interleaved
for pass in 1 2 3; do for c in 1 0; do for w in cyclegan cut hypelcnn; do
  echo "$w clock=$c pass $pass $(HYPEL_BENCH_CLOCK=$c timeout 300 python bench.py --workload $w --no-cpu-baseline --no-input-pipeline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done; done; done
