# round 5: the candidate loop's light table in LDS (LUM_LDS_LIGHTS) - parity, A/B against the build without it, and the batch size (ids per pass)
out=$1; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ambient_reuse.py -m gpu -x -q > $out/parity.log 2>&1; tail -2 $out/parity.log
WORKLOADS="hall scan example" bash tools/gpu_ab_variants.sh $out/ab_ldslights.txt nolds default nolds default
for spp in 64 128; do
  echo -n "hall, $spp ids per pass: " | tee -a $out/ab_ldslights.txt; python bench.py --steps 2 --warmup 1 --cpu-budget 0 --secondary none --workload hall --samples-per-pass $spp 2>/dev/null | python tools/ab_line.py | tee -a $out/ab_ldslights.txt
done
