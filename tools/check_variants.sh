#!/bin/bash
# Compiles every kept-but-off experiment of the ray kernels and of k_shade for gfx950 (device code only, nothing is linked or run), so that the variants
# behind macros do not rot unbuilt (advisor, round 5). Runs without a GPU: bash tools/check_variants.sh [jobs] > profiles/r06_variants_compile.txt
# A variant is a set of -D flags; it is compiled into the fast flavour's two translation units (wavefront_fast.hip, wavefront_fast_shadow.hip), and - where the
# host side takes part (node formats) - into core.hip, which also holds the exact flavour.
cd "$(dirname "$0")/.." || exit 1
JOBS=${1:-4}
HIPCC=${ROCM_PATH:-/opt/rocm}/bin/hipcc
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Os -fno-slp-vectorize --cuda-device-only -c -o /dev/null"
FAST="-DLUM_FAST=1 -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -freciprocal-math -fno-math-errno -fapprox-func -fgpu-flush-denormals-to-zero"
EXACT="-ffp-contract=off -fno-fast-math"
VARIANTS=(
  "phase_queues:-DLUM_PHASE_QUEUES=1:"
  "bvh8_octant:-DLUM_BVH8O=1:core"
  "bvh8_sorted:-DLUM_BVH8=1:core"
  "bvh4_quantised:-DLUM_BVH4Q=1:core"
  "speculate:-DLUM_SPECULATE=3:"
  "dual_visit:-DLUM_DUAL_VISIT=1:"
  "prefetch:-DLUM_PREFETCH=2:"
  "xcd_ranges:-DLUM_XCD_RANGES=1:"
  "defer_finish:-DLUM_DEFER_FINISH=1:"
  "lds_turn:-DLUM_LDS_TURN=16:"
  "lds_swizzle:-DLUM_LDS_SWIZZLE=1:core"
  "scalar_sobol:-DLUM_SCALAR_SOBOL=1:"
  "shade_staged:-DLUM_SHADE_STAGED=1:"
  "shade_static:-DLUM_SHADE_DYNAMIC=0:core"
  "phase_stats:-DLUM_PHASE_STATS:core"
  "leaf4:-DLUM_LEAF_MAX=4:core"
  "root_reference_form:-DLUM_ROOT_THRESHOLD=0:"
  "root_threshold_no_key:-DLUM_ROOT_KEY=0:"
  "root_integer_selects:-DLUM_ROOT_KEY=2:"
  "closest_all_slots:-DLUM_CLOSEST_ALL_SLOTS=1:"
  "prefetch_random:-DLUM_PREFETCH_RANDOM=1:"
  "dup_all:-DLUM_DUP=63:"
  "ablate_all:-DLUM_ABLATE=7 -DLUM_ABLATE_LIGHT=7:"
)
one() {
  local name=${1%%:*} rest=${1#*:}
  local flags=${rest%%:*} extra=${rest#*:}
  local ok=1 log
  for unit in "luminary_amd/csrc/device/wavefront_fast.hip $FAST -DLUM_SHADOW_KERNEL_EXTERN=1" "luminary_amd/csrc/device/wavefront_fast_shadow.hip $FAST" \
              ${extra:+"luminary_amd/csrc/host/core.hip $EXACT -DLUM_SHADOW_KERNEL_EXTERN=1"}; do
    # shellcheck disable=SC2086
    if ! log=$($HIPCC $COMMON $flags ${unit#* } ${unit%% *} 2>&1); then ok=0; echo "---- $name: ${unit%% *}"; echo "$log" | grep -E "error|Error" | head -5; fi
  done
  if [ $ok = 1 ]; then echo "[ok]     $name ($flags)"; else echo "[FAILED] $name ($flags)"; fi
}
export -f one; export HIPCC COMMON FAST EXACT
echo "# tools/check_variants.sh: off-by-default variants compiled for gfx950 (device code only), $(date -u +%Y-%m-%d), source $(git rev-parse --short HEAD 2>/dev/null)"
printf '%s\n' "${VARIANTS[@]}" | xargs -P "$JOBS" -I{} bash -c 'one "$@"' _ {}
