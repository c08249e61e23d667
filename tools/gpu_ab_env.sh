# A/B of runtime (environment) variants: bash tools/gpu_ab_env.sh "VAR=1" "VAR=2 OTHER=x" ... ; workloads in $WORKLOADS (default hall), extra bench args in $BENCH_ARGS
for e in "$@"; do
  for w in ${WORKLOADS:-hall}; do
    echo -n "[$e] $w $BENCH_ARGS: "
    env $e python bench.py --steps 3 --warmup 1 --cpu-budget 0 --secondary none --workload $w $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['config']['kernel_ms_rank0']; p=d["config"]["per_ray_rank0"]; print(round(d["value"],1),"Mrays/s trace %.1f shade %.1f shadow %.1f sort %.1f | nodes %.2f/%.2f tris %.2f/%.2f upload %.2fs" % (k["trace"], k["shade"], k["shadow"], k.get("sort", 0.0), p["nodes_closest"], p["nodes_shadow"], p["tris_closest"], p["tris_shadow"], d["config"]["scene_upload_s"]))"
  done
done
