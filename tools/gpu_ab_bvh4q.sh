#!/bin/bash
# Same-box A/B of the 64-byte quantised BVH4 nodes (LUM_BVH4Q) with parity checks of the variant: bash tools/gpu_ab_bvh4q.sh > gpurun_out/ab_bvh4q.txt
trap 'python -m luminary_amd.build --force > /dev/null 2>&1' EXIT
LUM_CXXFLAGS="-DLUM_BVH4Q=1" python -m luminary_amd.build --force > /dev/null 2>&1 || echo "build failed"
echo "== parity with LUM_BVH4Q=1"
LUM_CXXFLAGS="-DLUM_BVH4Q=1" timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_particles.py tests/test_lbvh.py -q -m gpu -x 2>&1 | tail -3
for rep in 1 2; do
for flags in "-DLUM_BVH4Q=0" "-DLUM_BVH4Q=1"; do
  LUM_CXXFLAGS="$flags" python -m luminary_amd.build --force > /dev/null 2>&1 || { echo "[$flags] build failed"; continue; }
  for w in hall scan example; do
    echo -n "[$flags] $w: "
    LUM_CXXFLAGS="$flags" python bench.py --steps 3 --warmup 1 --cpu-budget 0 --secondary none --workload $w 2>/dev/null | python tools/ab_line.py
  done
done
done
