// The wavefront kernels in the fast arithmetic flavour (flavour.h): this file is compiled with -DLUM_FAST=1 -ffp-contract=fast
// -fno-hip-fp32-correctly-rounded-divide-sqrt (luminary_amd/build.py) from the very same headers as the exact flavour in core.hip.
#if !defined(LUM_FAST) || !LUM_FAST
#error "wavefront_fast.hip is the fast flavour: build it with -DLUM_FAST=1"
#endif
#include "wavefront_table_impl.h"
