# round 5: the Sobol / Owen hash on the scalar unit where a wave's lanes share the sample id (LUM_SCALAR_SOBOL) - parity and A/B against the build without it
out=$1; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ambient_reuse.py tests/test_oracle_known_answers.py -m gpu -x -q > $out/parity.log 2>&1; tail -2 $out/parity.log
WORKLOADS="hall scan example" bash tools/gpu_ab_variants.sh $out/ab_sobol.txt nosobol default nosobol default
