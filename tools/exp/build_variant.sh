#!/bin/bash
# Builds a second copy of the library with extra compiler flags for a same-box A/B (HYPEL_LIB_PATH selects it):
#   tools/exp/build_variant.sh NAME "-DSOME_MACRO=0"   ->  hypelcnn_amd/csrc/variants/NAME/libhypel_hip.so
#   REV=<commit> tools/exp/build_variant.sh NAME          ->  seg_gemm.hip as of that commit (the other objects from the tree)
set -e
cd "$(dirname "$0")/../../hypelcnn_amd/csrc"
name=$1; shift
d=variants/$name
mkdir -p $d/build
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function $*"
src=seg_gemm.hip
if [ -n "$REV" ]; then git show $REV:hypelcnn_amd/csrc/seg_gemm.hip > $d/seg_gemm_rev.hip; src=$d/seg_gemm_rev.hip; fi
/opt/rocm/bin/hipcc $FL -c $src -o $d/build/seg_gemm.hip.o
objs=""
for s in elementwise.hip gan.hip gan_mfma.hip dense_stack.hip data.hip abi.cpp; do objs="$objs build/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libhypel_hip.so $d/build/seg_gemm.hip.o $objs
ls -la $d/libhypel_hip.so
