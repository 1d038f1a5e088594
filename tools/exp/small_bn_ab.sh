run() { printf "%-40s " "$*"; env "$@" python bench.py --steps 60 --no-cpu-baseline --no-input-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('%.3f ms  median %.3f  launches %d' % (d['ms_per_step'], d['ms_per_step_median'], r['kernel_launches_per_step']))"; }
for rep in 1 2; do run HYPEL_SMALL_BN=1; run HYPEL_SMALL_BN=0; done
