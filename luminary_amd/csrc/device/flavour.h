// Arithmetic flavour of the device code. The same headers are compiled twice into libluminary_amd.so:
//   exact  (LUM_FAST=0, -ffp-contract=off, correctly rounded / and sqrt, fixed polynomial sin/cos/atan2/exp2/log2): every sample is a
//          pure, bit-reproducible function of (scene, pixel, sample id); HIP == oracle bit for bit. All parity tests run this flavour.
//   fast   (LUM_FAST=1, -ffp-contract=fast, v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 / v_sin_f32 / v_exp_f32 / v_log_f32): what the reference
//          itself is built like (--use_fast_math, src/luminary/CMakeLists.txt:48). The product's default; gated by
//          tests/test_flavours.py (rel-L2 against `exact` < 1e-3 at 1024 spp, ray counters within 0.1 %).
// Everything flavoured lives in an inline namespace, so the two translation units' kernels and helpers get distinct symbols while the
// code keeps saying lum::x. Plain data layouts shared with the host side (dev_scene.h) stay in namespace lum itself.
#pragma once

#ifndef LUM_FAST
#define LUM_FAST 0
#endif
#if LUM_FAST
#define LUM_FLAVOUR_NS fast
#define LUM_FLAVOUR_NAME "fast"
#else
#define LUM_FLAVOUR_NS exact
#define LUM_FLAVOUR_NAME "exact"
#endif
#define LUM_NS_BEGIN namespace lum { inline namespace LUM_FLAVOUR_NS {
#define LUM_NS_END } }
