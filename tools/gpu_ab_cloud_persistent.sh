#!/bin/bash
# Same-box A/B of the persistent-lane cloud march (LUM_CLOUD_PERSISTENT), with parity of the default: bash tools/gpu_ab_cloud_persistent.sh > gpurun_out/ab_cloud_persistent.txt
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['config']['kernel_ms_rank0']; print('%.1f Mrays/s | sky group %.1f ms | %.1f ms/step' % (d['value'], k.get('sky',0.0), d['ms_per_step']))"; }
trap 'python -m luminary_amd.build --force > /dev/null 2>&1' EXIT
echo "== parity (default build)"; timeout 1500 python -m pytest tests/test_clouds.py -q -m gpu 2>&1 | tail -3
B="python bench.py --workload example --secondary none --cpu-budget 0 --steps 3 --warmup 1 --samples-per-pass 8"
for rep in 1 2; do
for flags in "-DLUM_CLOUD_PERSISTENT=0" "-DLUM_CLOUD_PERSISTENT=1"; do
  LUM_CXXFLAGS="$flags" python -m luminary_amd.build --force > /dev/null 2>&1 || { echo "[$flags] build failed"; continue; }
  for cfg in "--clouds" "--clouds --fog 40 --ocean 0.5"; do
    echo -n "[$flags] $cfg: "
    LUM_CXXFLAGS="$flags" $B $cfg 2>/dev/null | line
  done
done
done
