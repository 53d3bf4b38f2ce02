#!/bin/bash
# A/B build of the library with extra -D flags for kernels.hip:  dev/build_ab.sh NAME -DFLAG ...   -> dev/abx/libwaehip_NAME.so
# (run with WAE_LIB_PATH=dev/abx/libwaehip_NAME.so; dev/abx travels to the GPU box, its .so files are git-ignored)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../wavesandeigenvalues.jl_amd/csrc"
mkdir -p ../../dev/abx
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c kernels.hip -o /tmp/kernels_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../dev/abx/libwaehip_$NAME.so /tmp/kernels_$NAME.o lib.o amg.o assemble.o tiles.o mgpu.o -ldl
echo built dev/abx/libwaehip_$NAME.so
