#!/bin/bash
# shader clock and socket power while a workload of bench.py runs (rocm-smi sampled twice a second):
#   tools/gpu.sh --timeout 600 -- 'bash tools/exp/power_clock.sh hypelcnn 3000; bash tools/exp/power_clock.sh dualcnn 40'
w=${1:-hypelcnn}; steps=${2:-3000}
mkdir -p gpurun_out/power
python bench.py --workload $w --steps $steps --warmup 20 --no-cpu-baseline > gpurun_out/power/bench_$w.json 2> gpurun_out/power/bench_$w.err &
pid=$!
sleep 20   # import + planning + warm-up
for i in $(seq 1 ${SAMPLES:-16}); do
  kill -0 $pid 2>/dev/null || break
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
  sleep 0.5
done | tee gpurun_out/power/samples_$w.txt
wait $pid
cut -c1-160 gpurun_out/power/bench_$w.json
