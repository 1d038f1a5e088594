#!/bin/bash
# tools/exp/gan_kstats.sh <workload> [steps]: rocprofv3 kernel statistics of a GAN bench run -> per-step table
export TMPDIR=/tmp
ROOT=$(pwd); WL=${1:-cut}; STEPS=${2:-40}
rm -rf /tmp/gk_$WL
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gk_$WL -o t -- python $ROOT/bench.py --workload $WL --steps $STEPS --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$WL step ms', round(d['ms_per_step'],3), 'launches', d['roofline'].get('launches_per_step'))")
S=$(find /tmp/gk_$WL -name "*kernel_stats.csv" | head -1)
python tools/kstats.py $S 30
