#!/bin/bash
# Builds a second copy of the library with extra compiler flags for a same-box A/B (HYPEL_LIB_PATH selects it):
#   tools/exp/build_variant.sh NAME "-DSOME_MACRO=0"   ->  hypelcnn_amd/csrc/variants/NAME/libhypel_hip.so
#   REV=<commit> tools/exp/build_variant.sh NAME          ->  seg_gemm.hip as of that commit (the other objects from the tree)
#   FILE=dense_stack.hip tools/exp/build_variant.sh NAME -DDS_REGW=0   ->  the flags (or REV) apply to that source instead
set -e
cd "$(dirname "$0")/../../hypelcnn_amd/csrc"
name=$1; shift
d=variants/$name
mkdir -p $d/build
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function $*"
f=${FILE:-seg_gemm.hip}   # the source that gets the flags (or the revision); the other objects come from the tree's build/
src=$f
if [ -n "$REV" ]; then git show $REV:hypelcnn_amd/csrc/$f > $d/rev_$f; src=$d/rev_$f; fi
/opt/rocm/bin/hipcc $FL -c $src -o $d/build/$f.o
objs=""
for s in seg_gemm.hip elementwise.hip gan.hip gan_mfma.hip dense_stack.hip data.hip abi.cpp; do [ $s = $f ] || objs="$objs build/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libhypel_hip.so $d/build/$f.o $objs
ls -la $d/libhypel_hip.so
