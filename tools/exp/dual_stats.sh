#!/bin/bash
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/dual_stats; mkdir -p $OUT
W=${1:-dualcnn}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python $ROOT/bench.py --workload $W --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err
cd $ROOT
S=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
python - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print(f'{r["Name"][:100]:100s} {r["Calls"]:>6s} {float(r["TotalDurationNs"])/1e6:9.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}%')
PY
rm -rf $OUT/t
