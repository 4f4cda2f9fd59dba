#!/bin/bash
# Fixed (step-independent) part of one single-XCD rrLU launch at 685 x 688: kernel-trace durations for 1 / 16 / 64 / 256 pivot
# steps (GPU box, repo root).  Every profiler call is bounded; nothing here reads stdin.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for s in 1 16 64 256; do
  d=gpurun_out/fx$s
  timeout 150 rocprofv3 --kernel-trace --stats -d $d -o x --output-format csv -- python3 tools/probe_xcd.py child 685 688 $s ${1:-1} </dev/null >/dev/null 2>&1
  f=$d/x_kernel_stats.csv
  if [ -f "$f" ]; then echo "steps=$s $(grep rrlu_xcd "$f" </dev/null | cut -d, -f2-4,6,7)"; else echo "steps=$s: no stats"; fi
  rm -rf $d
done
