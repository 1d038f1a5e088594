#!/bin/bash
# Are any runtime blit / torch kernels launched PER STEP?  Trace the headline command at two step counts; a per-step kernel
# shows up as a slope in its call count (setup-only kernels have the same count in both runs).
#   tools/gpu.sh --timeout 900 -- 'bash tools/exp/blit_slope.sh'
export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out/blit
for n in 10 60; do
  cd /tmp; rm -rf /tmp/bl_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bl_$n -o t -- python $R/bench.py --steps $n --warmup 3 --no-cpu-baseline --no-input-pipeline > /dev/null 2> /tmp/bl_$n.err
  cd $R
  S=$(find /tmp/bl_$n -name "*kernel_stats.csv" | head -1)
  echo "== steps $n"
  python - "$S" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "rocclr" in n or "at::native" in n or "adam_tf1" in n or "copy_pair" in n or "fill_kernel" in n or n.startswith("fill"):
        print(f"   {n[:70]:70s} calls {r['Calls']}")
PY
done | tee gpurun_out/blit/slope.txt
