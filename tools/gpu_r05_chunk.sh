# rays a wave of the persistent ray kernels reserves per atomic (LUM_CHUNK_MAX, default 256)
out=gpurun_out/r05y; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  WORKLOADS="hall scan example" bash tools/gpu_ab_variants.sh $out/ab.txt default ch128 ch512 ch1024
done
