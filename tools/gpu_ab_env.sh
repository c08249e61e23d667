# A/B of runtime (environment) variants: bash tools/gpu_ab_env.sh "VAR=1" "VAR=2 OTHER=x" ... ; workloads in $WORKLOADS (default hall), extra bench args in $BENCH_ARGS
for e in "$@"; do
  for w in ${WORKLOADS:-hall}; do
    echo -n "[$e] $w $BENCH_ARGS: "
    env $e python bench.py --steps 3 --warmup 1 --cpu-budget 0 --secondary none --workload $w $BENCH_ARGS 2>/dev/null | python tools/ab_line.py
  done
done
