#!/bin/bash
# NEEDS tools/exp/bnsum_epilogue.patch applied (git apply; the form was measured neutral and is not in the tree) and the library rebuilt
# batch-norm backward sums in the data gradient's epilogue: kernel + model parity, then the step A/B on the same box
o=gpurun_out/r6_exp14; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "bnsum or res_epilogue or kslice" 2>&1 | tail -5 | tee $o/tests_kernels.txt
timeout 1500 python -m pytest tests/test_gpu_models.py -q -x -m gpu -k "hypelcnn or every_level" 2>&1 | tail -5 | tee $o/tests_models.txt
for pass in 1 2; do for set in "BNSUM_EPILOGUE=0" "BNSUM_EPILOGUE=1" "BNSUM_EPILOGUE=1,BNSUM_OVER_KSLICE=1"; do
  echo "$set pass $pass $(HYPEL_PLAN_SET=$set timeout 200 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-input-pipeline 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"].get("gemm_ms_per_step"))' 2>&1 | tail -1)"
done; done | tee $o/step_ab.txt
