#!/bin/bash
# what the clock sources of a box say while the two classifier workloads run
ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>/dev/null | head -40
for f in /sys/class/drm/card*/device/pp_dpm_sclk /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input /sys/class/drm/card*/device/hwmon/hwmon*/freq1_label /sys/class/drm/card*/device/hwmon/hwmon*/power1_average; do echo "== $f"; cat $f 2>&1 | head -12; done
for w in hypelcnn dualcnn; do
python bench.py --workload $w --no-cpu-baseline --no-input-pipeline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["workload"], d["ms_per_step"], {k:v for k,v in r.items() if "clock" in k or "power" in k})'
done
