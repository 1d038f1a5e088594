#!/bin/bash
# 1-rank RCCL data-parallel step: sync-point rule A/B (HYPEL_DP_SYNC_WORK), plus the single-device step
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 80 --warmup 20 --no-cpu-baseline --no-input-pipeline"
P='import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(sys.argv[1], round(d["ms_per_step"],4), round(d["ms_per_step_median"],4))'
for r in 1 2; do
for w in 0 0.5 0.3 0.7; do
  HYPEL_DP_SYNC_WORK=$w HYPEL_DP_SELFTEST=1 $RUN 2>/dev/null | python -c "$P" dp-work=$w
done; done
python bench.py --steps 80 --warmup 20 --no-cpu-baseline --no-input-pipeline 2>/dev/null | python -c "$P" single
