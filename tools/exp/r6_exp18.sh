#!/bin/bash
# NEEDS the experiment constant of tools/exp/merged_fwd_tile_order.patch (two lines of plan.py, not in the tree: the order lost)
# tile order of the merged forward inside a row chunk (L2 reuse of the packed weights / the chunk's pixel blocks): per launch and step
for pass in 1 2; do for o in 0 1 2; do
  echo "== order $o pass $pass"
  HYPEL_PLAN_SET=MERGE_FWD_ORDER=$o timeout 300 python tools/gemm_microbench.py --filter "fwd:connector" 2>&1 | grep "merged" | cut -c1-130
  echo "step order $o pass $pass $(HYPEL_PLAN_SET=MERGE_FWD_ORDER=$o timeout 200 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-input-pipeline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done; done
