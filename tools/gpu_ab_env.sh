# A/B of runtime (environment) variants: bash tools/gpu_ab_env.sh "VAR=1" "VAR=2 OTHER=x" ... ; extra bench args in $BENCH_ARGS
for e in "$@"; do
  echo "== [$e] $BENCH_ARGS"
  env $e python bench.py --steps 2 --warmup 1 --cpu-budget 0 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value'],1),'Mrays/s', d['config']['kernel_ms_rank0'], d['config']['per_ray_rank0'])"
done
