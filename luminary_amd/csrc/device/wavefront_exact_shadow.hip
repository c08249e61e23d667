// The exact flavour's visibility-ray kernel in a translation unit of its own, like the fast flavour's (wavefront_fast_shadow.hip): compiled with the max-ILP
// instruction scheduler (luminary_amd/build.py). The scheduler reorders instructions, it does not change one: the kernel stays bit-identical to the oracle
// (the whole GPU suite runs this flavour).
#if defined(LUM_FAST) && LUM_FAST
#error "wavefront_exact_shadow.hip is part of the exact flavour"
#endif
#if defined(LUM_SHADOW_KERNEL_EXTERN) && LUM_SHADOW_KERNEL_EXTERN
#error "this unit DEFINES k_shadow_rays"
#endif
#include <hip/hip_runtime.h>

#include "kernel_shadow.h"
