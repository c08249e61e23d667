# round 6, fourteenth call: what the random numbers' gathers cost k_shade (wrong images, timing only): rng1 = no blue-noise texel fetch, rng3 = no Sobol table word either
out=gpurun_out/r06n; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
WORKLOADS="hall example" bash tools/gpu_ab_variants.sh $out/ab.txt default rng1 rng3 default
