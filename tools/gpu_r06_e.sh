# round 6, fifth call: the light tree's root pass - threshold form in the fast flavour (default) against the reference's form (variant rform: -DLUM_ROOT_THRESHOLD=0), both with
# the upper-bound-only clamp; the parity and flavour tests first (the exact flavour's kernels changed by one instruction: v_min for v_med3)
out=gpurun_out/r06e; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_flavours.py tests/test_ambient_reuse.py tests/test_sobol_table.py -m gpu -q -x > $out/pytest_subset.log 2>&1; grep -E "passed|failed|error" $out/pytest_subset.log | tail -3
for rep in 1 2; do
  WORKLOADS="hall scan example" bash tools/gpu_ab_variants.sh $out/ab.txt default rform
done
