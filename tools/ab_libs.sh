# A/B of library builds on one GPU box: LIBS="suffix ..." (tensor4all-rs_amd/lib/libt4a_gpu<suffix>.so; "" = default)
for v in ${LIBS:-""}; do
  [ "$v" = "default" ] && v=""
  lib=$GRAFT_REPO_ROOT/tensor4all-rs_amd/lib/libt4a_gpu$v.so
  T4A_GPU_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --no-aux --steps 10 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['roofline']['latency_view']['us_per_pivot_step'])"
done
