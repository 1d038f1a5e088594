#!/bin/bash
# round 6, experiment 2: (A) merged level-1 forward on the split kernels: channel parts / taps per tile; (B) DUALCNN forward
# levels in weight-major tile order
o=gpurun_out/r6_exp2; mkdir -p $o
MB="python tools/gemm_microbench.py --rounds 8 --with-reduce --filter connector_"
i=0
for ps in "MERGE_FWD_MAX_COUT_SPLIT=32" "MERGE_FWD_MAX_COUT_SPLIT=32,L2_CHUNK_BYTES=33554432" \
          "MERGE_FWD_MAX_COUT_SPLIT=32,MERGE_MAX_TAPS=12" "MERGE_FWD_MAX_COUT_SPLIT=32,MERGE_MAX_TAPS=16" \
          "MERGE_FWD_MAX_COUT_SPLIT=32,MERGE_MAX_TAPS=25" "MERGE_FWD_MAX_COUT_SPLIT=32,MERGE_MAX_TAPS=16,L2_CHUNK_BYTES=33554432"; do
  HYPEL_PLAN_SET=$ps $MB > $o/a$i.txt 2>&1
  echo "== a$i $ps"; grep -h "fwd:connector_[12]\|tap-split" $o/a$i.txt | cut -c1-110
  i=$((i+1))
done
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],2), d["roofline"].get("gemm_ms_per_step"), d["roofline"].get("achieved"))'
for r in 1 2; do
for ps in "FWD_WEIGHT_MAJOR_BYTES=0" "FWD_WEIGHT_MAJOR_BYTES=16777216"; do
  HYPEL_PLAN_SET=$ps python bench.py --workload dualcnn --steps 10 --warmup 3 --no-cpu-baseline 2>$o/dual_err.txt | python -c "$P" "dualcnn $ps"
done
done
