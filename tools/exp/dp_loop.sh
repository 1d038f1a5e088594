#!/bin/bash
# run the RCCL self-test worker repeatedly; keep the log of the first failure
mkdir -p gpurun_out
for i in $(seq 1 ${1:-12}); do
  HYPEL_DP_SELFTEST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + i)) tests/dp_rccl_worker.py > gpurun_out/dp_loop.log 2>&1
  rc=$?
  if [ $rc -ne 0 ] || ! grep -q DP_RCCL_SELFTEST_OK gpurun_out/dp_loop.log; then
    echo "run $i FAILED rc=$rc"; grep -n "Error\|error\|what()\|terminate\|Traceback\|rank0\]:" gpurun_out/dp_loop.log | head -40; cp gpurun_out/dp_loop.log gpurun_out/dp_fail.log; exit 0
  fi
  echo "run $i ok"
done
