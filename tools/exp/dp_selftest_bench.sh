#!/bin/bash
# 1-rank RCCL self-test of the data-parallel bench path: plain DP step vs synchronised batch norm (same box)
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 60 --warmup 20 --no-cpu-baseline --no-input-pipeline"
P='import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(sys.argv[1], round(d["ms_per_step"],4), round(d["ms_per_step_median"],4), d["config"].get("batch_norm"))'
HYPEL_DP_SELFTEST=1 $RUN 2>/dev/null | python -c "$P" dp-local-bn
HYPEL_DP_SELFTEST=1 $RUN --sync-bn 2>/dev/null | python -c "$P" dp-sync-bn
python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-input-pipeline 2>/dev/null | python -c "$P" single
