#!/bin/bash
# MODE:SLEEP:extra-env   (each run: N iterations of [40 collectives, synchronize, capture 40 small graphs])
for cfg in "$@"; do
  m=$(echo $cfg | cut -d: -f1); s=$(echo $cfg | cut -d: -f2); e=$(echo $cfg | cut -d: -f3)
  env $e MODE=$m SLEEP=$s N=${N:-60} python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) tools/exp/capture_vs_watchdog.py > gpurun_out/cvw.log 2>&1
  echo "mode=$m sleep=$s env=$e rc=$? $(grep -c CAPTURE_LOOP_OK gpurun_out/cvw.log) ok; $(grep -m1 -o 'operation not permitted on an event last recorded in a capturing stream' gpurun_out/cvw.log)"
done
