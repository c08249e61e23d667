# where the batch-size effect comes from: kernel times per 3 steps at 16 / 32 / 64 / 128 sample ids per pass (hall), committed library
out=gpurun_out/r05p; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for b in 16 32 64 128; do
  BENCH_ARGS="--samples-per-pass $b" WORKLOADS="hall" bash tools/gpu_ab_variants.sh $out/ab.txt default
done
