# round 6, sixth call: the root pass with the picked child and its importance in one word per lane (variant rootkey: -DLUM_ROOT_KEY=1) against the default (threshold form, two words);
# which nodes count as the top of the tree (LUM_TOP_ORDER=area) with the new trees
out=gpurun_out/r06f; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  WORKLOADS="hall example" bash tools/gpu_ab_variants.sh $out/ab.txt default rootkey
  WORKLOADS="hall scan" bash tools/gpu_ab_env.sh "LUM_TOP_ORDER=area" | tee -a $out/ab.txt
done
LUM_LIB=$PWD/luminary_amd/lib/variants/rootkey/libluminary_amd.so timeout 900 python -m pytest tests/test_flavours.py -m gpu -q -x 2>&1 | tail -2
