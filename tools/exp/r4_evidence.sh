#!/bin/bash
# raw outputs of the round-4 experiments NOTES.md cites (same box, one call)
O=gpurun_out/r4_exp; mkdir -p $O
bash tools/exp/merge_sweep.sh > $O/level_merge_per_launch.txt 2>&1
bash tools/exp/merge_step_ab.sh > $O/level_merge_step_ab.txt 2>&1
bash tools/exp/astage_probe.sh > $O/astage_probe.txt 2>&1
bash tools/exp/small_launch_widths.sh > $O/small_launch_widths.txt 2>&1
bash tools/exp/chunk_bn_ab.sh > $O/chunk_bn_ab.txt 2>&1
bash tools/exp/small_bn_ab.sh > $O/small_bn_ab.txt 2>&1
bash tools/exp/wgrad_parallel_ab.sh > $O/wgrad_parallel_ab.txt 2>&1
{ for r in 1 2; do for t in 1 0; do for b in 1 0; do for wl in cut cyclegan; do
  [ $b = 0 ] && [ $t = 1 ] && continue
  HYPEL_GAN_GEN_TAP=$t HYPEL_GAN_BATCH_APPS=$b HYPEL_SLAB_REDUCE_MULTI=$b python bench.py --workload $wl --steps 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('batch_apps=$b tap=$t $wl', round(d['ms_per_step'],4), 'ms', r.get('launches_per_step'), 'launches  generator', round(r.get('generator_ms_per_step'),4), 'ms')"
done; done; done; done
for n in tree twotab; do if [ $n = tree ]; then L=""; else L="HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/alt/libhypel_$n.so"; fi
  env $L python bench.py --workload cut --steps 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('forward tap tables: $n (tree = one table)', round(d['ms_per_step'],4), 'ms  generator', round(r.get('generator_ms_per_step'),4))"; done; } > $O/gan_ab.txt 2>&1
ls -la $O
