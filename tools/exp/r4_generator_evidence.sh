#!/bin/bash
# tools/exp/r4_generator_evidence.sh (through tools/gpu.sh): raw outputs behind NOTES 4.J -- the generator kernels in isolation
# (register-ring backward on / off), their phase stamps and SQ counters, and the CUT / CycleGAN steps with the switch both ways.
# Needs alt builds: tools/exp/mkalt.sh gmdiag5 "-DGM_DIAG=5" gan_mfma.hip; tools/exp/mkalt.sh gmdiag7 "-DGM_DIAG=7" gan_mfma.hip
mkdir -p gpurun_out
ALT=$PWD/hypelcnn_amd/csrc/alt
{
echo "== tools/exp/gen_time.py, us per launch (in-tree library; HYPEL_GAN_BWD2=0 = the round-3 LDS-ring backward kernel)"
python tools/exp/gen_time.py
HYPEL_GAN_BWD2=0 GP_TAG=bwd2_off python tools/exp/gen_time.py
GP_B=64 GP_N=2048 python tools/exp/gen_time.py
HYPEL_GAN_BWD2=0 GP_TAG=bwd2_off GP_B=64 GP_N=2048 python tools/exp/gen_time.py
echo "== phase stamps of the register-ring backward kernel (-DGM_DIAG=5; 'forward recompute' = step A's element loop, 'write-out' = wave 0's own filter-gradient products)"
HYPEL_LIB_PATH=$ALT/libhypel_gmdiag5.so GP_KEPT=1 GP_N=4096 GP_B=360 python tools/exp/gen_phases.py
HYPEL_LIB_PATH=$ALT/libhypel_gmdiag5.so GP_KEPT=1 GP_N=2048 GP_B=64 python tools/exp/gen_phases.py
echo "== phase stamps of the forward kernel (-DGM_DIAG=7)"
HYPEL_LIB_PATH=$ALT/libhypel_gmdiag7.so python tools/exp/gen_fwd_phases.py
HYPEL_LIB_PATH=$ALT/libhypel_gmdiag7.so GP_N=2048 GP_B=64 python tools/exp/gen_fwd_phases.py
echo "== SQ counters of the generator kernels (tools/exp/gen_pmc.sh)"
bash tools/exp/gen_pmc.sh 2>&1 | grep -A1 "bwd2\|fwd_mfma_kernel<false, true, false>" | cut -c1-400
echo "== train steps, HYPEL_GAN_BWD2 = 1 / 0 (bench.py --steps 100 --no-cpu-baseline)"
for v in 1 0; do for wl in cut cyclegan; do
  HYPEL_GAN_BWD2=$v python bench.py --workload $wl --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('HYPEL_GAN_BWD2=$v', '$wl', 'ms/step', round(d['ms_per_step'],4), 'generator ms', round(r.get('generator_ms_per_step',0),4), 'frac', round(r['frac'],4))"
done; done
} 2>&1 | grep -v "amdgpu.ids\|simple_timer" > gpurun_out/r4_exp_generator_bwd2.txt
cat gpurun_out/r4_exp_generator_bwd2.txt
