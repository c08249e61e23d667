#!/bin/bash
# A/B of the feature kernels' register budgets on the Example-class scene (GPU box): bash tools/gpu_ab_features.sh > gpurun_out/ab_features.txt
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['config']['kernel_ms_rank0']; print('%.1f Mrays/s | trace %.1f shade %.1f shadow %.1f sky %.1f volume %.1f | %.1f ms/step' % (d['value'], k['trace'], k['shade'], k['shadow'], k.get('sky',0.0), k.get('volume',0.0), d['ms_per_step']))"; }
trap 'python -m luminary_amd.build --force > /dev/null 2>&1' EXIT
B="python bench.py --workload example --secondary none --cpu-budget 0 --steps 3 --warmup 1 --samples-per-pass 8"
for flags in "-DLUM_CLOUD_WAVES=2 -DLUM_FEATURE_WAVES=2" "-DLUM_CLOUD_WAVES=3 -DLUM_FEATURE_WAVES=3" "-DLUM_CLOUD_WAVES=4 -DLUM_FEATURE_WAVES=4"; do
  LUM_CXXFLAGS="$flags" python -m luminary_amd.build --force > /dev/null 2>&1 || { echo "[$flags] build failed"; continue; }
  for cfg in "--clouds" "--sky procedural --fog 40" "--sky procedural --ocean 0.5"; do
    echo -n "[$flags] $cfg: "
    LUM_CXXFLAGS="$flags" $B $cfg 2>/dev/null | line
  done
done
