#!/bin/bash
# A/B twins of the library for the single-XCD rrLU kernel (everything else from the default build):
#   libt4a_gpu_alt.so  default kernel with its phase stamps (-DT4A_XCD_STAMPS)
#   libt4a_gpu_p0.so   round-2 kernel (wave 0 is agent and poller: -DT4A_XCD_POLLER=0)
#   libt4a_gpu_p0s.so  round-2 kernel with stamps
# Select with T4A_GPU_LIB=<path>; stamps print with T4A_RRLU_STAMPS=1 (tools/probe_xcd.py).  Delete the files afterwards.
set -e
cd "$(dirname "$0")/../tensor4all-rs_amd"
python3 build.py > /dev/null
hipcc=${HIPCC:-/opt/rocm/bin/hipcc}
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -fvisibility=hidden"
others=$(ls build/*.o | grep -v kernels_rrlu_xcd.o)
build_one() { # name, extra flags
  $hipcc $F $2 -c csrc/kernels_rrlu_xcd.hip -o build/kernels_rrlu_xcd_$1.obj
  $hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libt4a_gpu_$1.so $others build/kernels_rrlu_xcd_$1.obj
  echo lib/libt4a_gpu_$1.so
}
build_one alt "-DT4A_XCD_STAMPS" &
build_one p0 "-DT4A_XCD_POLLER=0" &
build_one p0s "-DT4A_XCD_POLLER=0 -DT4A_XCD_STAMPS" &
wait
