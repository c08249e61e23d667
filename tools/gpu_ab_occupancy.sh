#!/bin/bash
# Same-box A/B of the ray kernels' occupancy: more, smaller workgroups per CU with a tighter register budget (bash tools/gpu_ab_occupancy.sh > gpurun_out/ab_occupancy.txt)
trap 'python -m luminary_amd.build --force > /dev/null 2>&1' EXIT
for flags in "-DLUM_TRACE_BLOCKS_PER_CU=1" "-DLUM_TRACE_BLOCK_FAST=512 -DLUM_TRACE_MIN_WAVES=4 -DLUM_TRACE_BLOCKS_PER_CU=2" "-DLUM_TRACE_BLOCK_FAST=640 -DLUM_TRACE_MIN_WAVES=5 -DLUM_TRACE_BLOCKS_PER_CU=2" "-DLUM_TRACE_BLOCK_FAST=768 -DLUM_TRACE_MIN_WAVES=6 -DLUM_TRACE_BLOCKS_PER_CU=2" "-DLUM_TRACE_BLOCKS_PER_CU=1"; do
  LUM_CXXFLAGS="$flags" python -m luminary_amd.build --force > /dev/null 2>&1 || { echo "[$flags] build failed"; continue; }
  for w in hall scan; do
    echo -n "[$flags] $w: "
    LUM_CXXFLAGS="$flags" python bench.py --steps 3 --warmup 1 --cpu-budget 0 --secondary none --workload $w 2>/dev/null | python tools/ab_line.py
  done
done
