{
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "generator or gan" 2>&1 | tail -3
for r in 1 2; do
python tools/exp/gen_time.py
HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/alt/libhypel_regskip0.so python tools/exp/gen_time.py
done
GP_B=64 GP_N=2048 python tools/exp/gen_time.py
HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/alt/libhypel_regskip0.so GP_B=64 GP_N=2048 python tools/exp/gen_time.py
} 2>&1 | grep -v amdgpu.ids
