# round 6, tenth call: what the parts of k_shade cost, each done TWICE (LUM_DUP, dev_light.h): results, paths and the other kernels stay what they are; the part's cost = the extra time
#   dup1 root pass   dup2 candidate loop   dup4 surface context   dup8 bounce sample   dup16 BSDF-driven light direction   dup32 local frame
out=gpurun_out/r06j; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
WORKLOADS="hall example" bash tools/gpu_ab_variants.sh $out/ab.txt default dup1 dup2 dup4 dup8 dup16 dup32 default
