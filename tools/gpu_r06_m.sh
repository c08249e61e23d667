# round 6, thirteenth call: more of the ray workgroups' LDS for the stacks (80 / 96 / 112 KB) on all three workloads
out=gpurun_out/r06m; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  WORKLOADS="hall scan example" bash tools/gpu_ab_variants.sh $out/ab.txt default stack80 stack96 stack112
done
