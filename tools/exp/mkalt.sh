#!/bin/bash
# tools/exp/mkalt.sh <name> "<extra flags>" [source.hip]  -> hypelcnn_amd/csrc/alt/libhypel_<name>.so
# (only the given source -- default seg_gemm.hip -- is rebuilt with the flags; the other objects come from build/)
set -e
cd "$(dirname "$0")/../../hypelcnn_amd/csrc"
SRC=${3:-seg_gemm.hip}
mkdir -p alt build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wall -Wno-unused-function $2 -c $SRC -o /tmp/alt_$1.o
OBJS=""
for s in seg_gemm.hip elementwise.hip gan.hip gan_mfma.hip dense_stack.hip data.hip abi.cpp; do
  if [ "$s" = "$SRC" ]; then OBJS="$OBJS /tmp/alt_$1.o"; else OBJS="$OBJS build/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o alt/libhypel_$1.so $OBJS
echo built alt/libhypel_$1.so
