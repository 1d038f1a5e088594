#!/bin/bash
# kernel statistics of the 1-rank RCCL data-parallel step vs the plain step (same box)
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/dp_trace; mkdir -p $OUT
cd /tmp
HYPEL_DP_SELFTEST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dp -o t -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-input-pipeline > $OUT/dp.json 2> $OUT/dp.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/single -o t -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-input-pipeline > $OUT/single.json 2> $OUT/single.err
cd $ROOT
for k in dp single; do
  S=$(find $OUT/$k -name "*kernel_stats.csv" | head -1)
  echo "== $k"; python - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f'{r["Name"][:90]:90s} {r["Calls"]:>6s} {float(r["TotalDurationNs"])/1e6:9.2f} ms')
PY
  grep -h "ms_per_step" $OUT/$k.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"
done
rm -rf $OUT/dp $OUT/single
