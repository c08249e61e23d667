# round 6, first GPU call: the GPU suite of the tree after the advisor fixes, smoke, then same-box A/Bs:
#   sample ids per pass 32 / 64 (hall, scan, example), and the 64-byte quantised nodes (variant bvh4q) re-measured against this round's kernels
out=gpurun_out/r06a; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for rep in 1 2; do
  for b in 32 64; do
    BENCH_ARGS="--samples-per-pass $b" WORKLOADS="hall scan example" bash tools/gpu_ab_variants.sh $out/ab.txt default
  done
  WORKLOADS="hall scan example" bash tools/gpu_ab_variants.sh $out/ab.txt bvh4q
done
