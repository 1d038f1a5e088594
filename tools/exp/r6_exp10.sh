#!/bin/bash
# round 6, experiment 10: DUALCNN with merged data / filter gradients of its biased levels; CUT per-launch table
o=gpurun_out/r6_exp10; mkdir -p $o
python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "dualcnn" > $o/tests_dual.txt 2>&1; tail -3 $o/tests_dual.txt
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],2), d["roofline"].get("gemm_ms_per_step"), d["roofline"].get("achieved"))'
for r in 1 2; do
  HYPEL_MERGE_LEVELS=0 python bench.py --workload dualcnn --steps 10 --warmup 3 --no-cpu-baseline 2>$o/err.txt | python -c "$P" "dualcnn unmerged"
  python bench.py --workload dualcnn --steps 10 --warmup 3 --no-cpu-baseline 2>$o/err.txt | python -c "$P" "dualcnn merged<=64"
  HYPEL_PLAN_SET=MERGE_LEVELS_MAX_COUT=480,MERGE_WGRAD_MAX_COUT=480 python bench.py --workload dualcnn --steps 10 --warmup 3 --no-cpu-baseline 2>$o/err480.txt | python -c "$P" "dualcnn merged<=480"
  HYPEL_MERGE_LEVELS=wgrad python bench.py --workload dualcnn --steps 10 --warmup 3 --no-cpu-baseline 2>$o/err.txt | python -c "$P" "dualcnn merged wgrad only"
done
python tools/exp/gan_launch_table.py --workload cut > $o/cut_launch_table.txt 2>&1; tail -8 $o/cut_launch_table.txt | cut -c1-400
HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/variants/clk2/libhypel_hip.so python tools/exp/drain_probe.py > $o/drain_idle_capacity.txt 2>&1; tail -30 $o/drain_idle_capacity.txt
HYPEL_PLAN_SET=KSLICE=0 HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/variants/clk2/libhypel_hip.so python tools/exp/drain_probe.py > $o/drain_idle_capacity_nokslice.txt 2>&1; tail -3 $o/drain_idle_capacity_nokslice.txt
