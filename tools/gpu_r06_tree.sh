# round 6: what the 4-wide collapse rule and the leaf size are worth on the GPU (host builder, so that both rules exist; the GPU SAH builder makes the same binary trees)
#   greedy = the rule of rounds 1-5 (largest child opened until four), optimal = the dynamic programme over the binary tree (bvh_build.cpp CollapsePlan)
#   leaf2 library = -DLUM_LEAF_MAX=2: leaves of at most two triangles AND leaf registers for two (24 VGPRs instead of 48)
out=gpurun_out/r06b; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
L2=$PWD/luminary_amd/lib/variants/leaf2/libluminary_amd.so
for rep in 1 2; do
  WORKLOADS="hall scan example" bash tools/gpu_ab_env.sh "LUM_BVH_BUILDER=sah LUM_BVH_COLLAPSE=0" "LUM_BVH_BUILDER=sah LUM_BVH_COLLAPSE=1" "LUM_BVH_BUILDER=sah LUM_BVH_COLLAPSE=1 LUM_BVH_MAX_LEAF=2" \
     "LUM_LIB=$L2 LUM_BVH_BUILDER=sah LUM_BVH_COLLAPSE=0" "LUM_LIB=$L2 LUM_BVH_BUILDER=sah LUM_BVH_COLLAPSE=1" | tee -a $out/ab.txt
done
