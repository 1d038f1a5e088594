#!/bin/bash
# tools/exp/step_kernels.sh <grep pattern> name...  : rocprofv3 kernel statistics of the headline bench for alt libraries
# ("tree" = the in-tree build), only the kernels matching the pattern; LAST="<name> <n>": the last n launches of a kernel
export TMPDIR=/tmp
ROOT=$(pwd)
PAT=$1; shift
for n in "$@"; do
  rm -rf /tmp/sp_$n
  if [ "$n" = tree ]; then unset HYPEL_LIB_PATH; else export HYPEL_LIB_PATH=$ROOT/hypelcnn_amd/csrc/alt/libhypel_$n.so; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_$n -o t -- python $ROOT/bench.py --workload ${WORKLOAD:-hypelcnn} --steps 30 --warmup 10 --no-cpu-baseline --no-input-pipeline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n step ms', round(d['ms_per_step'],3))")
  S=$(find /tmp/sp_$n -name "*kernel_stats.csv" | head -1)
  python tools/kstats.py $S 60 | grep -i -E "$PAT" | sed "s/^/$n  /"
  if [ -n "$LAST" ]; then python tools/exp/last_launches.py $(find /tmp/sp_$n -name "*kernel_trace.csv" | head -1) $LAST | sed "s/^/$n  /"; fi
done
