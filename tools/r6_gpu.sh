#!/bin/bash
# usage (on the GPU box, through gpurun): tools/r6_gpu.sh STAGE...   — one parametrised launcher for the round-6 measurements.
# Every stage writes under gpurun_out/r6_<stage>/.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
for stage in "$@"; do
  out=gpurun_out/r6_$stage
  mkdir -p "$out"
  case $stage in
    small)      # small-problem engine: parity tests + configs[1] time to solution
      timeout 900 python -m pytest tests/test_gpu_small.py -x -q > "$out/pytest.log" 2>&1; echo "pytest rc $?" >> "$out/pytest.log"; tail -15 "$out/pytest.log"
      timeout 300 python tools/probe_cfg2.py > "$out/probe_cfg2.log" 2>&1; cat "$out/probe_cfg2.log" ;;
    tci)        # every TCI2 / chain test
      timeout 2400 python -m pytest tests/test_gpu_small.py tests/test_gpu_tci2.py tests/test_gpu_chain.py -x -q > "$out/pytest.log" 2>&1; echo "pytest rc $?" >> "$out/pytest.log"; tail -15 "$out/pytest.log" ;;
    suite)      # the whole GPU suite
      timeout 3400 python -m pytest tests -m gpu -x -q > "$out/pytest.log" 2>&1; echo "pytest rc $?" >> "$out/pytest.log"; tail -15 "$out/pytest.log" ;;
    rng)        # the streams changed in round 6 (tree proposers, ACI initial guess), ADVICE fixes, the new bench modes
      timeout 3000 python -m pytest tests/test_gpu_bench_cli.py tests/test_gpu_tree.py tests/test_gpu_aci.py tests/test_gpu_quantics.py tests/test_gpu_tci2.py tests/test_gpu_rook.py tests/test_gpu_parallel.py tests/test_gpu_pishard.py -x -q > "$out/pytest.log" 2>&1; echo "pytest rc $?" >> "$out/pytest.log"; tail -15 "$out/pytest.log" ;;
    modes)      # N = 1 numbers of the two multi-GPU modes
      timeout 900 python bench.py --mode patch-farm --steps 5 --warmup 1 > "$out/patch_farm.json" 2> "$out/patch_farm.err"; tail -2 "$out/patch_farm.json"
      timeout 900 python bench.py --mode pi-shard --steps 2 --warmup 1 > "$out/pi_shard.json" 2> "$out/pi_shard.err"; tail -2 "$out/pi_shard.json" ;;
    ab)         # A/B of a variant library against the default one: AB_LIB=tensor4all-rs_amd/lib/libt4a_gpu_<name>.so tools/r6_gpu.sh ab
      for arm in default variant default variant; do
        if [ $arm = variant ]; then export T4A_GPU_LIB=$PWD/${AB_LIB}; else unset T4A_GPU_LIB; fi
        timeout 600 python bench.py --no-aux --no-components --no-cpu-baseline --steps 20 --warmup 3 2>> "$out/ab.err" | python -c "import sys, json; d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$arm', round(d['ms_per_step'], 3), 'ms per sweep,', round(d['roofline']['latency_view']['us_per_pivot_step'], 4), 'us per pivot step')" | tee -a "$out/ab.log"
      done
      export T4A_GPU_LIB=$PWD/${AB_LIB}
      timeout 1200 python -m pytest tests/test_gpu_dense.py tests/test_gpu_fuzz.py -x -q -k "rrlu or luci" > "$out/variant_pytest.log" 2>&1; echo "variant pytest rc $?" | tee -a "$out/ab.log"; tail -3 "$out/variant_pytest.log"
      unset T4A_GPU_LIB ;;
    bench)      # the default bench line
      timeout 900 python bench.py > "$out/bench.json" 2> "$out/bench.err"; tail -3 "$out/bench.json" ;;
    components) timeout 900 python tools/bench_components.py > "$out/components.json" 2> "$out/components.err"; tail -5 "$out/components.json" ;;
    *) echo "unknown stage $stage" ;;
  esac
done
