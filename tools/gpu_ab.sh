# A/B of compile-time variants on the GPU box: bash tools/gpu_ab.sh "<flags 1>" "<flags 2>" ...   (workloads in $WORKLOADS, default hall scan;
# extra bench arguments in $BENCH_ARGS, e.g. "--flavour exact"). The default build is restored on exit, whatever happens.
trap 'python -m luminary_amd.build --force > /dev/null 2>&1' EXIT
for flags in "$@"; do
  LUM_CXXFLAGS="$flags" python -m luminary_amd.build --force > /dev/null 2>&1 || { echo "[$flags] build failed"; continue; }
  for w in ${WORKLOADS:-hall scan}; do
    echo -n "[$flags] $w $BENCH_ARGS: "
    LUM_CXXFLAGS="$flags" python bench.py --steps 3 --warmup 1 --cpu-budget 0 --secondary none --workload $w $BENCH_ARGS 2>/dev/null | python tools/ab_line.py
  done
done
