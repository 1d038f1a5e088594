{
for r in 1 2 3 4 5; do python bench.py --workload cyclegan --steps 300 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cyc ms', round(d['ms_per_step'],4), 'median', round(d.get('ms_per_step_median',0),4), d.get('ms_per_step_p10_p90'))"; done
} 2>&1 | grep -v amdgpu.ids
