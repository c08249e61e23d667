# k_light_query compiled for more waves per SIMD (LUM_LQ_WAVES; the grid follows): a latency-bound kernel (wait 0.86) at 4 waves of 111 registers
out=gpurun_out/r05t; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  WORKLOADS="hall scan example" bash tools/gpu_ab_variants.sh $out/ab.txt default lq5 lq6 lq8
done
