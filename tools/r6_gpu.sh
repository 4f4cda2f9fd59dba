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
    profiles)   # the round's profile artefacts (copy the summaries from gpurun_out/r6_profiles into profiles/r06_*)
      export T4A_ROUND=r06
      B="python3 bench.py --no-cpu-baseline --no-aux --no-floor"
      prof() { timeout -k 5 400 rocprofv3 "$@" </dev/null; }
      timeout 900 python3 bench.py --steps 10 --warmup 3 2> "$out/bench_n1.err" | tail -1 > "$out/bench_n1.json"
      prof --kernel-trace --stats -d "$out/stats" -o x --output-format csv -- $B --steps 10 --warmup 3 > "$out/stats.log" 2>&1
      cp "$out/stats/x_kernel_stats.csv" "$out/bench_n1_kernel_stats.csv"
      prof --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch" -o x --output-format csv -- $B > "$out/pmc_fetch.log" 2>&1
      prof --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_write" -o x --output-format csv -- $B > "$out/pmc_write.log" 2>&1
      python3 tools/pmc_summary.py "$out/pmc_fetch" "$out/pmc_write" > "$out/pmc_summary.log" 2>&1
      cp profiles/r06_pmc_* "$out/" 2>/dev/null
      prof --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_MFMA -d "$out/pmc_mfma" -o x --output-format csv -- $B > "$out/pmc_mfma.log" 2>&1
      python3 tools/pmc_sq_summary.py "$out/pmc_mfma" "$out/pmc_mfma_per_kernel.csv" "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_MFMA -- $B" > /dev/null
      python3 tools/mfma_summary.py "$out/pmc_mfma_per_kernel.csv" "$out/bench_n1_kernel_stats.csv" "$out/mfma_kernels.json" > /dev/null 2>&1
      # the small-problem engine under the profiler: kernel durations of configs[1] (one launch per solve)
      prof --kernel-trace --stats -d "$out/cfg2stats" -o x --output-format csv -- python3 tools/probe_cfg2.py > "$out/cfg2_probe.log" 2>&1
      grep -i "small_optimize\|chain_walk\|Name" "$out/cfg2stats/x_kernel_stats.csv" > "$out/cfg2_kernel_stats.csv"
      timeout 300 python3 tools/probe_cfg2.py > "$out/cfg2_probe_unprofiled.log" 2>&1
      timeout 300 python3 tools/probe_fill.py > "$out/fill_probe.txt" 2>&1
      timeout 300 python3 tools/probe_linalg.py > "$out/linalg_probe.txt" 2>&1
      timeout 600 python3 tools/probe_cfg4_variants.py > "$out/cfg4_variants.txt" 2>&1
      timeout 900 python3 bench.py --mode patch-farm --steps 5 --warmup 1 2>/dev/null | tail -1 > "$out/bench_patch_farm_n1.json"
      timeout 900 python3 bench.py --mode pi-shard --steps 2 --warmup 1 2>/dev/null | tail -1 > "$out/bench_pi_shard_n1.json"
      timeout 600 python3 bench.py --mode site-shard --steps 3 --warmup 1 2>/dev/null | tail -1 > "$out/bench_site_shard_n1.json"
      timeout 1500 python3 tools/bench_components.py > "$out/components.json" 2> "$out/components.err"
      rm -rf "$out/stats" "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_mfma" "$out/cfg2stats"
      ls -la "$out" | head -40 ;;
    bench)      # the default bench line
      timeout 900 python bench.py > "$out/bench.json" 2> "$out/bench.err"; tail -3 "$out/bench.json" ;;
    components) timeout 900 python tools/bench_components.py > "$out/components.json" 2> "$out/components.err"; tail -5 "$out/components.json" ;;
    *) echo "unknown stage $stage" ;;
  esac
done
