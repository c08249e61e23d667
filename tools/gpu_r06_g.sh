# round 6, seventh call: the root pass's selects as integer instructions (variant rootkey2: v_ashrrev_i32 / v_bfi_b32 / v_min_u32, no comparison, no scalar mask) against rootkey and the default
out=gpurun_out/r06g; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  WORKLOADS="hall example" bash tools/gpu_ab_variants.sh $out/ab.txt default rootkey rootkey2
done
LUM_LIB=$PWD/luminary_amd/lib/variants/rootkey2/libluminary_amd.so timeout 900 python -m pytest tests/test_flavours.py -m gpu -q -x 2>&1 | tail -2
