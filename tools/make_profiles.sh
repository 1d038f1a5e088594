#!/bin/bash
# Regenerates the round's profile evidence on the GPU box (run through gpurun from the repo root):
#   tools/make_profiles.sh <tag> [workloads]     e.g. r3 "hypelcnn dualcnn cut cyclegan"  (default: all four)
#   -> gpurun_out/prof_<tag>/<files>; copy the summaries into profiles/ (tools/collect_profiles.sh <tag>)
# per workload:
#   bench line (driver contract)                        -> bench_<wl>.json
#   rocprofv3 --kernel-trace --stats of the same cmd    -> kernel_stats_<wl>.csv (+ per_launch_<wl>.txt for classifiers)
#   classifiers: PMC passes FETCH_SIZE / WRITE_SIZE     -> hbm_traffic_<wl>.{txt,json}, hbm_traffic_per_launch_<wl>.txt
#   headline: PMC pass of the SQ counters               -> mfma_busy_per_launch_hypelcnn.txt
set -u
TAG=${1:-r4}
WLS=${2:-"hypelcnn dualcnn cut cyclegan"}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
for WL in $WLS; do
  case $WL in
    hypelcnn) STEPS=100; TSTEPS=20; KNOWN=$((1024*49*145*4)); NB=1024;;
    dualcnn)  STEPS=30;  TSTEPS=4;  KNOWN=$((512*121*49*4));  NB=512;;
    *)        STEPS=100; TSTEPS=30; KNOWN=0; NB=0;;
  esac
  EXTRA=""; [ $WL = hypelcnn ] || EXTRA="--workload $WL"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_$WL -o t -- python $ROOT/bench.py $EXTRA --steps $TSTEPS --warmup 3 --no-cpu-baseline --no-input-pipeline > $ROOT/$OUT/trace_bench_$WL.json 2> $ROOT/$OUT/trace_$WL.err
  cd $ROOT
  S=$(find $OUT/trace_$WL -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/kernel_stats_$WL.csv
  T=$(find $OUT/trace_$WL -name "*kernel_trace.csv" | head -1)
  if [ $NB -gt 0 ]; then
    [ -n "$T" ] && python profiles/analyze_trace.py $T --workload $WL > $OUT/per_launch_$WL.txt 2>&1
    cd /tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/pmc_f_$WL -o f -- python $ROOT/bench.py $EXTRA --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-input-pipeline > /dev/null 2> $ROOT/$OUT/pmc_f_$WL.err
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/$OUT/pmc_w_$WL -o w -- python $ROOT/bench.py $EXTRA --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-input-pipeline > /dev/null 2> $ROOT/$OUT/pmc_w_$WL.err
    cd $ROOT
    F=$(find $OUT/pmc_f_$WL -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_w_$WL -name "*counter_collection.csv" | head -1)
    [ -n "$F" ] && [ -n "$W" ] && python tools/pmc_traffic.py $F $W --workload $WL --batch $NB --known-bytes $KNOWN --json $OUT/hbm_traffic_$WL.json --source "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py $EXTRA --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-input-pipeline (every step of the run incl. pre-warm and event replay), tools/pmc_traffic.py" > $OUT/hbm_traffic_$WL.txt 2>&1
    [ -n "$F" ] && [ -n "$W" ] && python tools/pmc_traffic_per_launch.py $F $W --workload $WL > $OUT/hbm_traffic_per_launch_$WL.txt 2>&1
    rm -rf $OUT/pmc_f_$WL $OUT/pmc_w_$WL
    # the bench line below quotes roofline.traffic from profiles/<tag>_hbm_traffic*.json: put THIS call's passes there first
    if [ -s $OUT/hbm_traffic_$WL.json ]; then
      if [ $WL = hypelcnn ]; then cp $OUT/hbm_traffic_$WL.json profiles/${TAG}_hbm_traffic.json; else cp $OUT/hbm_traffic_$WL.json profiles/${TAG}_hbm_traffic_$WL.json; fi
    fi
    if [ $WL = hypelcnn ]; then   # matrix-core utilisation per launch from the SQ counters (own pass, no tracing)
      cd /tmp
      rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $ROOT/$OUT/pmc_sq_$WL -o q -- python $ROOT/bench.py $EXTRA --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-input-pipeline > /dev/null 2> $ROOT/$OUT/pmc_sq_$WL.err
      cd $ROOT
      Q=$(find $OUT/pmc_sq_$WL -name "*counter_collection.csv" | head -1)
      [ -n "$Q" ] && python tools/pmc_sq_per_launch.py $Q --workload $WL > $OUT/mfma_busy_per_launch_$WL.txt 2>&1
      [ -s $OUT/mfma_busy_per_launch_$WL.txt ] && cp $OUT/mfma_busy_per_launch_$WL.txt profiles/${TAG}_mfma_busy_per_launch_$WL.txt
      rm -rf $OUT/pmc_sq_$WL
    fi
  else
    [ -n "$S" ] && python tools/kstats.py $S 30 > $OUT/kernel_top_$WL.txt 2>&1
  fi
  rm -rf $OUT/trace_$WL   # raw traces are large; the summaries above are what gets committed
  # the driver-contract line, after the PMC passes (its roofline.traffic is this call's)
  python bench.py $EXTRA --steps $STEPS > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
  # the same step with the RCCL path active on a 1-rank communicator (the only multi-process check a 1-GPU box allows):
  # what the data-parallel machinery itself costs (sync-point graph cuts, all-reduce launches, the flag exchange)
  HYPEL_DP_SELFTEST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py $EXTRA --steps $STEPS --no-cpu-baseline --no-input-pipeline > $OUT/dp_selftest_$WL.json 2> $OUT/dp_selftest_$WL.err
done
# rehearsal of the bare multi-GPU command (bench.py starts its own ranks): two gloo ranks sharing this one MI355X
HYPEL_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-input-pipeline 2> $OUT/bench_selflaunch_gloo2.err | tail -1 > $OUT/bench_selflaunch_gloo2.json
python - <<PY > $OUT/dp_selftest_overhead.txt
import json, glob, os
print("1-rank RCCL self-test (HYPEL_DP_SELFTEST=1 under torch.distributed.run --nproc-per-node 1) vs the plain run, same box")
for wl in "$WLS".split():
    try:
        a = json.loads(open("$OUT/bench_%s.json" % wl).read().strip().splitlines()[-1])
        b = json.loads(open("$OUT/dp_selftest_%s.json" % wl).read().strip().splitlines()[-1])
        print("%-9s plain %.4f ms/step   1-rank RCCL %.4f ms/step   %+.2f %%" % (wl, a["ms_per_step"], b["ms_per_step"], 100 * (b["ms_per_step"] / a["ms_per_step"] - 1)))
    except Exception as e:
        print(wl, "missing:", e)
PY
ls -la $OUT
# every bench line of the set must be one valid JSON line: a failed leg (e.g. the 2-rank self-launch rehearsal) must not pass unseen
python - <<PY | tee $OUT/validity.txt
import glob, json
bad = 0
for f in sorted(glob.glob("$OUT/bench_*.json") + glob.glob("$OUT/dp_selftest_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("ok     ", f, d.get("n_gpus"), round(d["ms_per_step"], 4))
    except Exception as e:
        bad += 1
        print("INVALID", f, repr(e)[:80])
print("PROFILE SET", "COMPLETE" if not bad else "HAS %d INVALID BENCH LINES" % bad)
PY
