#!/bin/bash
# tools/exp/hetero_ab.sh: CycleGAN / CUT / DCL steps with and without two-variable-set launches (HYPEL_GAN_BATCH_HETERO)
for rep in 1 2; do
for wl in cyclegan cut; do
for h in 1 0; do
  HYPEL_GAN_BATCH_HETERO=$h python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl hetero=$h step ms', round(d['ms_per_step'],4), 'launches', d['roofline'].get('launches_per_step'), 'gen frac', d['roofline'].get('frac'))"
done; done; done
