cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_sq -o q -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-input-pipeline > /dev/null 2> /tmp/sq.err
tail -3 /tmp/sq.err
Q=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1)
cd $GRAFT_REPO_ROOT && head -3 $Q && python tools/pmc_sq_per_launch.py $Q | tee gpurun_out/mfma_busy_per_launch_hypelcnn.txt | tail -50
