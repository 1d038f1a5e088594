#!/bin/bash
# prize of folding the batch-norm backward reduction into the producing data gradient's epilogue: the step without those launches
for pass in 1 2; do for d in "" bn_act_bwd_reduce bn_act_bwd_reduce,bwd_reduce_finalize; do
  echo "drop=[$d] pass $pass $(DROP=$d timeout 200 python tools/exp/drop_probe.py --steps 100 --warmup 30 --no-cpu-baseline --no-input-pipeline 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])' 2>&1 | tail -1)"
done; done
