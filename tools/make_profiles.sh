#!/bin/bash
# Regenerates the round's profile evidence on the GPU box (run through gpurun from the repo root):
#   tools/make_profiles.sh <tag>      e.g. r2   -> gpurun_out/prof_<tag>/...; copy the summaries into profiles/
# 1. bench line (driver contract)                      -> bench_n1.json
# 2. rocprofv3 --kernel-trace --stats of the same cmd  -> kernel_stats.csv, per_launch.txt (profiles/analyze_trace.py)
# 3. PMC passes FETCH_SIZE / WRITE_SIZE (separate)      -> hbm_traffic.{txt,json}
set -u
TAG=${1:-r2}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace -o t -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-input-pipeline > $ROOT/$OUT/trace_bench.json 2> $ROOT/$OUT/trace.err
cd $ROOT
S=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/kernel_stats.csv
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python profiles/analyze_trace.py $T > $OUT/per_launch.txt 2>&1
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/pmc_f -o f -- python $ROOT/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --no-input-pipeline > /dev/null 2> $ROOT/$OUT/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/$OUT/pmc_w -o w -- python $ROOT/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --no-input-pipeline > /dev/null 2> $ROOT/$OUT/pmc_w.err
cd $ROOT
F=$(find $OUT/pmc_f -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_w -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_traffic.py $F $W --json $OUT/hbm_traffic.json --source "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --no-input-pipeline (every step of the run incl. pre-warm and event replay), tools/pmc_traffic.py" > $OUT/hbm_traffic.txt 2>&1
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_traffic_per_launch.py $F $W > $OUT/hbm_traffic_per_launch.txt 2>&1
rm -rf $OUT/trace $OUT/pmc_f $OUT/pmc_w   # raw traces are large; the summaries above are what gets committed
ls -la $OUT
