"""bench.py as the driver runs it: a child process, one JSON line on stdout (the contract of the task description)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_single_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--no-aux"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["rccl_world_size"] == 1 and line["steps"] == 1 and line["warmup"] == 0
    assert line["unit"] == "GF/s" and line["dtype"] == "f64" and line["value"] > 0 and line["ms_per_step"] > 0
    assert "chi_max=256" in line["config"]["workload"] and line["config"]["n_sites"] == 30
    roof = line["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and 0 < roof["frac"] < 1 and roof["achieved"] > 0
    assert "mfma" in roof, "the MFMA kernels of the sweep are part of the line (BASELINE.json metric: ... + MFMA %)"


def test_bench_refuses_a_world_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=2" in (out.stderr + out.stdout)


def test_bench_site_shard_mode_single_gpu():
    """`--mode site-shard` (BASELINE.json configs[3]: d = 40, chi = 512) at N = 1: runs to its JSON line — round 4 found this mode
    faulting in 25 - 75 % of its runs (replay of the captured fill graph on a handle with asynchronous core export) — and the line
    says what is replicated, what shards and the bound that follows."""
    for _ in range(2):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "site-shard", "--gpus", "1", "--steps", "1",
                              "--warmup", "1"], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        line = json.loads(lines[0])
        assert line["n_gpus"] == 1 and line["config"]["chi_max"] == 512 and line["config"]["n_sites"] == 40
        am = line["amdahl"]
        assert am["replicated_ms"] > 0 and am["sharded_ms"] > 0 and am["speedup_bound_vs_n1"] >= 1.0
        assert line["scaling"] == "strong"


def _one_line(argv, timeout=900):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_patch_farm_mode_single_gpu():
    """`--mode patch-farm` (BASELINE.json configs[4] as stated: 64 statically projected patches, chi = 128, PaddedPatchFarm +
    DevicePatchExporter(group = 8), one all_gather_into_tensor) at N = 1: one JSON line with the per-rank times, the gathered bytes
    and the exposed gather time; every patch reaches chi.  No scaling claim at N = 1."""
    line = _one_line(["--mode", "patch-farm", "--gpus", "1", "--steps", "1", "--warmup", "1"])
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["unit"] == "patches/s"
    assert line["config"]["n_patches"] == 64 and line["config"]["chi_max"] == 128 and line["config"]["group"] == 8
    assert line["all_patches_reach_chi"] is True
    assert len(line["per_rank_ms_per_step"]) == 1 and line["per_rank_ms_per_step"][0] > 0
    assert line["gather_bytes_per_step"] >= 64 * 30 * 128 * 2 * 128 * 8
    assert 0 <= line["exposed_gather_ms_per_step"][0] < line["ms_per_step"]
    assert abs(line["value"] - 64 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]


def test_bench_pi_shard_mode_single_gpu():
    """`--mode pi-shard` (candidate matrices of a host callback by column blocks, rrLU replicated) at N = 1: the
    unsharded callback path with its three costs side by side — the user function, the (absent) gather, the rrLU kernels."""
    line = _one_line(["--mode", "pi-shard", "--gpus", "1", "--steps", "1", "--warmup", "1"])
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["config"]["chi_max"] == 256
    assert line["callback_ms_per_step"][0] > 0 and line["rrlu_ms_per_step"][0] > 0 and line["gather_ms_per_step"][0] == 0.0
    assert line["callback_points_per_step_all_ranks"] > 1e7
    # (callback_ms is an ESTIMATE: points x the per-point cost calibrated on a probe batch; it is most of the step, and noise of the probe
    # may push it a few per cent past the measured wall time)
    assert 0.5 * line["ms_per_step"] < line["callback_ms_per_step"][0] < 1.25 * line["ms_per_step"]
