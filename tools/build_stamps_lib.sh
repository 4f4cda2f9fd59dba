#!/bin/bash
# Diagnostic twin of the library: both generations of the single-XCD rrLU kernel and the one-wave kernel compiled with their per-phase cycle stamps
# (-DT4A_XCD_STAMPS), every other object taken from the default build.  Select it with
# T4A_GPU_LIB=<repo>/tensor4all-rs_amd/lib/libt4a_gpu_alt.so and T4A_RRLU_STAMPS=1 (tools/probe_xcd.py prints the stamps).
# STAMP_WAVE=<w> stamps wave w of rank 0 instead of the polling wave (second generation only); STAMP_EXTRA="-D..." STAMP_NAME=<suffix>: a
# variant of the kernels under the stamps.  Delete the files afterwards.
set -e
cd "$(dirname "$0")/../tensor4all-rs_amd"
python3 build.py > /dev/null
hipcc=${HIPCC:-/opt/rocm/bin/hipcc}
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -fvisibility=hidden -DT4A_XCD_STAMPS -DT4A_XCD_STAMP_WAVE=${STAMP_WAVE:-0} ${STAMP_EXTRA:-}"
OUT=lib/libt4a_gpu_alt${STAMP_WAVE:+_w$STAMP_WAVE}${STAMP_NAME:+_$STAMP_NAME}.so
$hipcc $FL -c csrc/kernels_rrlu_xcd.hip -o build/kernels_rrlu_xcd_stamps.obj &
$hipcc $FL -c csrc/kernels_rrlu_xcd2.hip -o build/kernels_rrlu_xcd2_stamps${STAMP_WAVE:+_w$STAMP_WAVE}.obj &
$hipcc $FL -Iinclude -I../include -c csrc/kernels_rrlu_w1.hip -o build/kernels_rrlu_w1_stamps.obj &
$hipcc $FL -c csrc/kernels_rrlu_xcd2m.hip -o build/kernels_rrlu_xcd2m_stamps${STAMP_WAVE:+_w$STAMP_WAVE}.obj &
wait
$hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $(ls build/*.o | grep -v "kernels_rrlu_xcd.o\|kernels_rrlu_xcd2.o\|kernels_rrlu_w1.o\|kernels_rrlu_xcd2m.o") build/kernels_rrlu_xcd2m_stamps${STAMP_WAVE:+_w$STAMP_WAVE}.obj build/kernels_rrlu_w1_stamps.obj build/kernels_rrlu_xcd_stamps.obj build/kernels_rrlu_xcd2_stamps${STAMP_WAVE:+_w$STAMP_WAVE}.obj
echo $OUT
