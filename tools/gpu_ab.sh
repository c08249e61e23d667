# A/B of compile-time variants on the GPU box: bash tools/gpu_ab.sh "<flags 1>" "<flags 2>" ...   (workloads in $WORKLOADS, default both)
for flags in "$@"; do
  LUM_CXXFLAGS="$flags" python -m luminary_amd.build --force > /dev/null 2>&1
  for w in ${WORKLOADS:-example hall}; do
    echo -n "[$flags] $w: "
    python bench.py --steps 3 --warmup 1 --cpu-budget 0 --workload $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['config']['kernel_ms_rank0']; print(round(d['value'],1),'Mrays/s trace %.1f shade %.1f shadow %.1f' % (k['trace'], k['shade'], k['shadow']))"
  done
done
