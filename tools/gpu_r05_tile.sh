# camera rays generated in t x t pixel tiles instead of pixel rows (bench.py --pixel-tile: a pixel list in tile order), committed library
out=gpurun_out/r05z; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for t in 0 8 16 32; do
    BENCH_ARGS="--pixel-tile $t" WORKLOADS="hall scan example" bash tools/gpu_ab_variants.sh $out/ab.txt default
  done
done
