# round 6, eighth call: the tree with the root pass's key and the one-v_rsq importance (fast flavour) - full GPU suite, then default against rootkey (= the same without the rsq change)
out=gpurun_out/r06h; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $out/pytest_gpu.log | tail -3
for rep in 1 2; do
  WORKLOADS="hall example scan" bash tools/gpu_ab_variants.sh $out/ab.txt default rootkey
done
