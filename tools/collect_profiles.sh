#!/bin/bash
# tools/collect_profiles.sh <tag>: copy the summaries of gpurun_out/prof_<tag>/ into profiles/<tag>_* (tracked)
TAG=${1:-r4}
SRC=gpurun_out/prof_$TAG
for f in $SRC/validity.txt $SRC/bench_*.json $SRC/kernel_stats_*.csv $SRC/per_launch_*.txt $SRC/hbm_traffic_*.txt $SRC/hbm_traffic_*.json $SRC/hbm_traffic_per_launch_*.txt $SRC/mfma_busy_per_launch_*.txt $SRC/kernel_top_*.txt $SRC/dp_selftest_overhead.txt; do
  [ -s "$f" ] || continue
  b=$(basename $f)
  cp $f profiles/${TAG}_$b
done
# bench.py's roofline.traffic reads profiles/<tag>_hbm_traffic.json (headline) and <tag>_hbm_traffic_dualcnn.json
[ -s profiles/${TAG}_hbm_traffic_hypelcnn.json ] && mv profiles/${TAG}_hbm_traffic_hypelcnn.json profiles/${TAG}_hbm_traffic.json
ls profiles/${TAG}_*
