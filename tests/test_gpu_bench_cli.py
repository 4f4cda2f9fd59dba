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
