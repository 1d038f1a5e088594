#!/bin/bash
# STAT_BLOCKS (chunks per 64-column stripe of the batch-norm reduction passes) -- same-box A/B on the headline step
#   tools/gpu.sh --timeout 1200 -- 'bash tools/exp/stat_blocks_ab.sh'
mkdir -p gpurun_out/statblocks
for pass in 1 2; do for v in 256 512 1024; do
  export HYPEL_PLAN_SET="STAT_BLOCKS=$v"
  echo "STAT_BLOCKS=$v pass $pass: $(python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-input-pipeline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'ms  median', round(d['ms_per_step_median'],4))")"
done; done
for v in 256 1024; do
  export HYPEL_PLAN_SET="STAT_BLOCKS=$v"
  echo "== STAT_BLOCKS=$v per kernel (eager, event pairs)"
  python tools/exp/nongemm_roofline.py 2>/dev/null | grep -E "^(bn_act_bwd_reduce|bwd_reduce_finalize|bn_finalize|col_stats_partial|bn_act_fwd|bn_act_bwd_apply) +[0-9]+ "
done
