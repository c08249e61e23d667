#!/bin/bash
# Same-box A/B of the hit compaction in k_ocean_shade / k_particle_shade, with parity of the default: bash tools/gpu_ab_hit_compact.sh > gpurun_out/ab_hit_compact.txt
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['config']['kernel_ms_rank0']; print('%.1f Mrays/s | shade group %.1f ms | %.1f ms/step' % (d['value'], k['shade'], d['ms_per_step']))"; }
trap 'python -m luminary_amd.build --force > /dev/null 2>&1' EXIT
echo "== (parity of the default build: run separately)"
B="python bench.py --workload example --secondary none --cpu-budget 0 --steps 3 --warmup 1 --samples-per-pass 8"
for rep in 1 2; do
for flags in "-DLUM_HIT_COMPACT=0" "-DLUM_HIT_COMPACT=1"; do
  LUM_CXXFLAGS="$flags" python -m luminary_amd.build --force > /dev/null 2>&1 || { echo "[$flags] build failed"; continue; }
  for cfg in "--sky procedural --ocean 0.5" "--sky procedural --ocean 6"; do
    echo -n "[$flags] $cfg: "
    LUM_CXXFLAGS="$flags" $B $cfg 2>/dev/null | line
  done
done
done
