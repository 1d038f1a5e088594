python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for lib in default head default head; do
  if [ $lib = default ]; then unset HYPEL_LIB_PATH; else export HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/variants/$lib/libhypel_hip.so; fi
  echo "== $lib"
  python bench.py --workload dualcnn --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['achieved'])"
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['achieved'])"
done
