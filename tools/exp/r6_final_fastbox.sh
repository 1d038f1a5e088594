#!/bin/bash
# the round's profile call, on a box that is not one of the pool's slow ones: a 40 s calibration first (the same binary spreads
# 5.53 - 5.79 ms over boxes this round; CUT, whose code did not change, spreads 1.25 - 1.33 ms with it), then tools/exp/r6_final.sh
o=gpurun_out/r6_final; mkdir -p $o
ms=$(python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-input-pipeline 2>/dev/null | python -c 'import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])["ms_per_step"])')
echo "calibration: $ms ms/step (limit ${LIMIT:-5.62})" | tee $o/calibration.txt
if python -c "import sys; sys.exit(0 if float('$ms') <= float('${LIMIT:-5.62}') else 1)"; then
  bash tools/exp/r6_final.sh
else
  echo "SLOW BOX: profiles not taken"
fi
