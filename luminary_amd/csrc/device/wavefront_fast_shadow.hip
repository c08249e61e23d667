// The fast flavour's visibility-ray kernel in a translation unit of its own, so that it - and only it - is compiled with the max-ILP instruction scheduler
// (luminary_amd/build.py adds `-mllvm -amdgpu-sched-strategy=max-ilp` to this file; kernel_shadow.h has the measurements). Same headers, same flags otherwise.
#if !defined(LUM_FAST) || !LUM_FAST
#error "wavefront_fast_shadow.hip is part of the fast flavour: build it with -DLUM_FAST=1"
#endif
#if defined(LUM_SHADOW_KERNEL_EXTERN) && LUM_SHADOW_KERNEL_EXTERN
#error "this unit DEFINES k_shadow_rays"
#endif
#include <hip/hip_runtime.h>

#include "kernel_shadow.h"
