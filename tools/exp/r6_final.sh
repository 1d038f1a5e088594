#!/bin/bash
# full GPU suite, smoke, then the round's profiles
o=gpurun_out/r6_final; mkdir -p $o
python -m pytest tests -x -q -m gpu > $o/gputests.txt 2>&1; tail -3 $o/gputests.txt
python __graft_entry__.py smoke > $o/smoke.txt 2>&1; tail -2 $o/smoke.txt
bash tools/make_profiles.sh r6 > $o/make_profiles.log 2>&1; tail -5 $o/make_profiles.log
for w in hypelcnn dualcnn cut cyclegan; do tail -c 600 gpurun_out/prof_r6/bench_$w.json; echo; done
