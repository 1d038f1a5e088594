#!/bin/bash
# round 6, experiment 11: do the filter gradients fill the drains of the data-gradient chain on a second (low-priority) stream?
o=gpurun_out/r6_exp11; mkdir -p $o
python tools/exp/overlap_probe.py 2>&1 | grep -v amdgpu.ids | tee $o/overlap_default.txt
MAIN_HI=1 python tools/exp/overlap_probe.py 2>&1 | grep -v amdgpu.ids | tee $o/overlap_main_hi.txt
