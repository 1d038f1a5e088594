#!/bin/bash
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 80 --warmup 20 --no-cpu-baseline --no-input-pipeline"
P='import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(sys.argv[1], round(d["ms_per_step"],4), round(d["ms_per_step_median"],4))'
for r in 1 2; do
HYPEL_DP_SELFTEST=1 $RUN 2>/dev/null | python -c "$P" dp-marker
HYPEL_DP_MARKER=0 HYPEL_DP_SELFTEST=1 $RUN 2>/dev/null | python -c "$P" dp-nomarker
HYPEL_DP_NO_SYNC=1 HYPEL_DP_SELFTEST=1 $RUN 2>/dev/null | python -c "$P" dp-nosync
python bench.py --steps 80 --warmup 20 --no-cpu-baseline --no-input-pipeline 2>/dev/null | python -c "$P" single
done
