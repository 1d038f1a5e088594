#!/bin/bash
# Every launch of one train step with its rocprofv3 duration, in launch order -> gpurun_out/trace_all.txt
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/trace_all
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-input-pipeline > $OUT/bench.json 2> $OUT/trace.err
cd $ROOT
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python profiles/analyze_trace.py $T all > gpurun_out/trace_all.txt 2>&1
rm -rf $OUT/trace
