#!/bin/bash
# A/B twin of the library: the second-generation single-XCD rrLU kernel (solo + group translation units) compiled with extra
# flags, every other object taken from the default build.  usage: tools/build_variant_lib.sh NAME "-DFLAG=.. ..."
# Select it with T4A_GPU_LIB=<repo>/tensor4all-rs_amd/lib/libt4a_gpu_NAME.so.  Delete the files afterwards (git-ignored).
set -e
NAME=$1
shift
cd "$(dirname "$0")/../tensor4all-rs_amd"
hipcc=${HIPCC:-/opt/rocm/bin/hipcc}
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -fvisibility=hidden $*"
$hipcc $FL -c csrc/kernels_rrlu_xcd2.hip -o build/kernels_rrlu_xcd2_$NAME.obj &
$hipcc $FL -c csrc/kernels_rrlu_xcd2_group.hip -o build/kernels_rrlu_xcd2_group_$NAME.obj &
wait
$hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libt4a_gpu_$NAME.so $(ls build/*.o | grep -v "kernels_rrlu_xcd2.o\|kernels_rrlu_xcd2_group.o") build/kernels_rrlu_xcd2_$NAME.obj build/kernels_rrlu_xcd2_group_$NAME.obj
echo lib/libt4a_gpu_$NAME.so
