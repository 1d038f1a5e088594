{
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "generator or gan" 2>&1 | tail -3
python tools/exp/gen_time.py
GP_B=64 GP_N=2048 python tools/exp/gen_time.py
python bench.py --workload cut --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cut bwd2 ms', d['ms_per_step'], d['roofline']['frac'], d['roofline']['generator_ms_per_step'])"
python bench.py --workload cyclegan --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cyc bwd2 ms', d['ms_per_step'])"
} 2>&1 | grep -v amdgpu.ids
