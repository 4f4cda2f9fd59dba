#!/bin/bash
# round-4 A/B session: headline bench (ms per cfg3 sweep) for a list of configurations "LIBNAME[:ENV=VAL[,ENV=VAL]]" (LIBNAME =
# default or the NAME of lib/libt4a_gpu_NAME.so), two interleaved repetitions; digests of the default library against the
# chip-wide kernel; phase stamps of the polling wave and of wave 3.    usage: tools/r4_gpu_ab.sh OUTDIR [--pytest] CONFIG...
O=gpurun_out/$1
shift
mkdir -p $O
export PYTHONPATH=tensor4all-rs_amd/python
L=$PWD/tensor4all-rs_amd/lib
if [ "$1" == "--pytest" ]; then
  shift
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
  echo "pytest rc=$?" >> $O/pytest.log
  tail -2 $O/pytest.log
fi
echo "== probe default" > $O/probe.log
timeout 600 python tools/probe_xcd.py 685 688 256 512 512 256 100 700 100 130 130 130 >> $O/probe.log 2>&1
for rep in 1 2; do
for cfg in "$@"; do
  lib=${cfg%%:*}
  envs=""
  [ "$cfg" != "$lib" ] && envs=$(echo "${cfg#*:}" | tr ',' ' ')
  tagname=$(echo "$cfg" | tr ':=,' '___')
  if [ "$lib" == "default" ]; then libenv=""; else libenv="T4A_GPU_LIB=$L/libt4a_gpu_$lib.so"; fi
  env $libenv $envs timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-aux > $O/bench_${tagname}_$rep.json 2> $O/bench_$tagname.err
done
done
echo "== stamps" > $O/stamps.log
for lib in alt alt_w3; do
[ -f $L/libt4a_gpu_$lib.so ] && T4A_RRLU_STAMPS=1 T4A_GPU_LIB=$L/libt4a_gpu_$lib.so timeout 300 python tools/probe_xcd.py child 685 688 256 1 2>&1 | tail -3 >> $O/stamps.log
done
grep -c digest $O/probe.log
for f in $O/bench_*.json; do echo -n "$f: "; python -c "import json,sys; d=json.load(open('$f')); print(round(d['ms_per_step'],3), round(d['value'],1))"; done
