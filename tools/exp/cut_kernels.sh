#!/bin/bash
# tools/exp/cut_kernels.sh name...  : rocprofv3 kernel statistics of the CUT bench for alt libraries (generator kernels only)
export TMPDIR=/tmp
ROOT=$(pwd)
for n in "$@"; do
  rm -rf /tmp/cp_$n
  (cd /tmp && HYPEL_LIB_PATH=$ROOT/hypelcnn_amd/csrc/alt/libhypel_$n.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_$n -o t -- python $ROOT/bench.py --workload ${WORKLOAD:-cut} --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n step ms', round(d['ms_per_step'],3))")
  S=$(find /tmp/cp_$n -name "*kernel_stats.csv" | head -1)
  python tools/kstats.py $S 40 | grep -i "generator\|nce" | sed "s/^/$n  /"
done
