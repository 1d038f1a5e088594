#!/bin/bash
# full GPU suite + the headline bench (two passes) + microbench of the data gradients
o=gpurun_out/r6_full; mkdir -p $o
python -m pytest tests -x -q -m gpu > $o/gputests.txt 2>&1; tail -3 $o/gputests.txt
python tools/gemm_microbench.py --rounds 8 --with-reduce --filter dgrad > $o/mb_dgrad.txt 2>&1; grep -h "dgrad\|kslice\|TOTAL" $o/mb_dgrad.txt | grep -v "fc_\|image_gen\|conv_2 \|connector_conv_[12]" | cut -c1-118
BENCH="python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-input-pipeline"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), round(d["ms_per_step_median"],4), round(d["roofline"]["gemm_ms_per_step"],4))'
for r in 1 2; do
  $BENCH 2>/dev/null | python -c "$P" default
  HYPEL_PLAN_SET=KSLICE=0 $BENCH 2>/dev/null | python -c "$P" kslice-off
  HYPEL_PLAN_SET=KSLICE_FRAC_MIN=0.25 $BENCH 2>/dev/null | python -c "$P" kslice-0.25
done
python bench.py > $o/bench_default.json 2>$o/bench_err.txt; tail -c 1500 $o/bench_default.json
