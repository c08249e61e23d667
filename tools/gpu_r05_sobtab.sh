# the pass's Sobol / Owen table (dev_sampler.h SamplerT<kTable>, lumc_set_sobol_table): its own tests, the parity of the table path in the exact flavour, then A/B
#   default: k_shade<.., kTable> instances chosen per pass; the same library with LUM_SOBOL_TABLE_RT=0 hashes (the kTable = false instances)
out=gpurun_out/r05n; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_sobol_table.py tests/test_gpu_parity.py tests/test_ambient_reuse.py tests/test_flavours.py -m gpu -x -q 2>&1 | tail -3 | tee $out/parity3.txt
for rep in 1 2; do
  WORKLOADS="hall scan example" bash tools/gpu_ab_variants.sh $out/ab3.txt default
  echo "[default, LUM_SOBOL_TABLE_RT=0]" | tee -a $out/ab3.txt
  LUM_SOBOL_TABLE_RT=0 WORKLOADS="hall scan example" bash tools/gpu_ab_variants.sh $out/ab3.txt default
done
