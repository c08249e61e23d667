# the ended paths' vertices resolved by the next depth's k_shade (fused_flags & 4) instead of k_resolve_ended: parity, then A/B on one library (LUM_FUSED_ENDED=0: the kernel)
out=gpurun_out/r05r; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_ambient_reuse.py tests/test_sobol_table.py tests/test_flavours.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee $out/parity.txt
for rep in 1 2; do
  echo "[k_resolve_ended, LUM_FUSED_ENDED=0]" | tee -a $out/ab.txt
  LUM_FUSED_ENDED=0 WORKLOADS="hall scan example" bash tools/gpu_ab_variants.sh $out/ab.txt default
  echo "[in the next depth's k_shade]" | tee -a $out/ab.txt
  WORKLOADS="hall scan example" bash tools/gpu_ab_variants.sh $out/ab.txt default
done
