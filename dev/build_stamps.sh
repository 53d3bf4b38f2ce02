#!/bin/bash
# diagnostic build of the library with in-kernel s_memtime stamps in spmv_tile_kernel (dev/tile_stamps.py reads them)
set -e
cd "$(dirname "$0")/../wavesandeigenvalues.jl_amd/csrc"
mkdir -p ../../dev/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DWAE_TILE_STAMPS $EXTRA -c kernels.hip -o /tmp/kernels_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../dev/ab/${OUT:-libwaehip_stamps.so} /tmp/kernels_stamps.o lib.o amg.o assemble.o tiles.o mgpu.o -ldl
