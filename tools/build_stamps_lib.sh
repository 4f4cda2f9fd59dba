#!/bin/bash
# Diagnostic twin of the library: the single-XCD rrLU kernel compiled with its per-phase cycle stamps (-DT4A_XCD_STAMPS), every
# other object taken from the default build.  Select it with T4A_GPU_LIB=<repo>/tensor4all-rs_amd/lib/libt4a_gpu_alt.so and
# T4A_RRLU_STAMPS=1 (tools/probe_xcd.py prints the stamps).  Delete the file afterwards.
set -e
cd "$(dirname "$0")/../tensor4all-rs_amd"
python3 build.py > /dev/null
hipcc=${HIPCC:-/opt/rocm/bin/hipcc}
$hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -fvisibility=hidden -DT4A_XCD_STAMPS -c csrc/kernels_rrlu_xcd.hip -o build/kernels_rrlu_xcd_stamps.obj
$hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libt4a_gpu_alt.so $(ls build/*.o | grep -v kernels_rrlu_xcd.o) build/kernels_rrlu_xcd_stamps.obj
echo lib/libt4a_gpu_alt.so
