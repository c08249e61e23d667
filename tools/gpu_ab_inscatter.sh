#!/bin/bash
# Same-box A/B of the path compaction in k_volume_inscatter: bash tools/gpu_ab_inscatter.sh > gpurun_out/ab_inscatter.txt
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['config']['kernel_ms_rank0']; print('%.1f Mrays/s | volume group %.1f ms | %.1f ms/step' % (d['value'], k['volume'], d['ms_per_step']))"; }
trap 'python -m luminary_amd.build --force > /dev/null 2>&1' EXIT
echo "== parity (default build)"; timeout 2000 python -m pytest tests/test_ocean.py tests/test_fog.py -q -m gpu 2>&1 | tail -2
B="python bench.py --workload example --secondary none --cpu-budget 0 --steps 3 --warmup 1 --samples-per-pass 8"
for rep in 1 2; do
for flags in "-DLUM_INSCATTER_COMPACT=0" "-DLUM_INSCATTER_COMPACT=1"; do
  LUM_CXXFLAGS="$flags" python -m luminary_amd.build --force > /dev/null 2>&1 || { echo "[$flags] build failed"; continue; }
  for cfg in "--sky procedural --ocean 0.5" "--sky procedural --ocean 6" "--sky procedural --fog 40"; do
    echo -n "[$flags] $cfg: "
    LUM_CXXFLAGS="$flags" $B $cfg 2>/dev/null | line
  done
done
done
