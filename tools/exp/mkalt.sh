#!/bin/bash
# mkalt.sh name "extra flags"  -> alt/libhypel_<name>.so (only seg_gemm.hip rebuilt with the flags)
set -e
cd /root/repo/hypelcnn_amd/csrc
mkdir -p alt build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wall -Wno-unused-function $2 -c seg_gemm.hip -o /tmp/alt_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o alt/libhypel_$1.so /tmp/alt_$1.o build/elementwise.hip.o build/gan.hip.o build/data.hip.o build/abi.cpp.o
echo built alt/libhypel_$1.so
