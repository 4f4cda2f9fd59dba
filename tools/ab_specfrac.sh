# A/B of the speculation threshold of the single-XCD kernel (fraction of the previous pivot's square), GPU box
for v in ${SPECS:-0.66 0.75 0.85 0.95 1.1 0.66}; do
  T4A_XCD_SPECFRAC=$v timeout 200 python bench.py --no-cpu-baseline --no-aux --steps 10 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['roofline']['latency_view']['us_per_pivot_step'])"
done
