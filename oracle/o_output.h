/*
 * ORACLE (test infrastructure, not product): output chain, moments -> display-referred RGB -> ARGB8.
 *
 * Restated from cuda/accumulation.cuh:86-153 (result = first moment / sample count), cuda/kernels.cuh:503-556
 * (generate_final_image), cuda/tonemap.cuh:7-246, cuda/purkinje.cuh:19-90, cuda/math.cuh:1044-1060 (sRGB), :1081-1168 (filters),
 * :1483-1543 (HSV), cuda/post_common.cuh:6-44 (bilinear fetch), cuda/kernels.cuh:558-644 (convert_RGBF_to_ARGB8),
 * cuda/random.cuh:150-154, :197-212, :305-307, :370-379 (dither / grain masks).
 * Scope: any supersampling and undersampling stage (incl. accumulation_generate_result_undersampling, accumulation.cuh:192-254) and bloom (device/device_post.c).
 * The reference's log2f/powf/rsqrtf are fast-math approximations with unspecified bits; log2/exp2/pow are fixed polynomial
 * sequences here (the HIP path uses the same sequences), which keeps the image bytes reproducible. Parity unpinned, like the
 * rest of the oracle: no golden image exists in the reference.
 */
#ifndef ORACLE_O_OUTPUT_H
#define ORACLE_O_OUTPUT_H

#include "o_rng.h"

typedef struct {
  uint32_t src_width, src_height, dst_width, dst_height;
  float inv_sample_count, exposure;
  uint32_t tonemap, filter, dithering, purkinje, use_color_correction, passthrough;
  float purkinje_kappa1, purkinje_kappa2;
  float cc_h, cc_s, cc_v;
  float film_grain;
  float agx_slope, agx_power, agx_saturation;
  uint32_t supersampling, undersampling_stage; /* src = nominal output size << supersampling; stage s > 0: the input is the compact (src >> s) image */
} OracleOutputParams;

static inline float o_linear_to_srgb(float v) { return (v <= 0.0031308f) ? 12.92f * v : 1.055f * o_pow(v, 0.416666666667f) - 0.055f; }
static inline float o_srgb_to_linear(float v) { return (v <= 0.04045f) ? v / 12.92f : o_pow((v + 0.055f) / 1.055f, 2.4f); }

/* tonemap.cuh:7-36 */
static inline RGBF o_tonemap_aces(RGBF px) {
  RGBF c = c3(0.59719f * px.r + 0.35458f * px.g + 0.04823f * px.b, 0.07600f * px.r + 0.90834f * px.g + 0.01566f * px.b,
              0.02840f * px.r + 0.13383f * px.g + 0.83777f * px.b);
  RGBF a = c_add(c, c_splat(0.0245786f));
  a = c_mul(c, a);
  a = c_add(a, c_splat(-0.000090537f));
  RGBF b = c_mul(c, c_splat(0.983729f));
  b = c_add(b, c_splat(0.432951f));
  b = c_mul(c, b);
  b = c_add(b, c_splat(0.238081f));
  b = c3(1.0f / b.r, 1.0f / b.g, 1.0f / b.b);
  c = c_mul(a, b);
  return c3(1.60475f * c.r - 0.53108f * c.g - 0.07367f * c.b, -0.10208f * c.r + 1.10813f * c.g - 0.00605f * c.b,
            -0.00327f * c.r - 0.07276f * c.g + 1.07602f * c.b);
}
/* tonemap.cuh:38-63 */
static inline float o_uncharted2_partial(float v) {
  const float a = 0.15f, b = 0.50f, c = 0.10f, d = 0.20f, e = 0.02f, f = 0.30f;
  return ((v * (a * v + c * b) + d * e) / (v * (a * v + b) + d * f)) - e / f;
}
static inline RGBF o_tonemap_uncharted2(RGBF px) {
  px = c_mul(px, c_splat(2.0f));
  px = c3(o_uncharted2_partial(px.r), o_uncharted2_partial(px.g), o_uncharted2_partial(px.b));
  const float s = 1.0f / o_uncharted2_partial(11.2f);
  return c_mul(px, c_splat(s));
}
static inline RGBF o_tonemap_reinhard(RGBF px) { const float f = 1.0f / (1.0f + c_luminance(px)); return c3(px.r * f, px.g * f, px.b * f); }
/* tonemap.cuh:80-176 */
static inline float o_agx_contrast(float v) {
  const float v2 = v * v, v4 = v2 * v2;
  return 15.5f * v4 * v2 - 40.14f * v4 * v + 31.96f * v4 - 6.868f * v2 * v + 0.4298f * v2 + 0.1191f * v - 0.00232f;
}
static inline float o_agx_encode(float v) {
  const float lo = -12.47393f, hi = 4.026069f;
  v = fmaxf(v, 0.00017578139f);
  v = fminf(fmaxf(o_log2(v), lo), hi);
  return o_agx_contrast((v - lo) / (hi - lo));
}
static inline RGBF o_agx_conversion(RGBF px) {
  RGBF a = c_splat(0.0f);
  a = c_add(a, c_scale(c3(0.842479062253094f, 0.0423282422610123f, 0.0423756549057051f), px.r));
  a = c_add(a, c_scale(c3(0.0784335999999992f, 0.878468636469772f, 0.0784336f), px.g));
  a = c_add(a, c_scale(c3(0.0792237451477643f, 0.0791661274605434f, 0.879142973793104f), px.b));
  return c3(o_agx_encode(a.r), o_agx_encode(a.g), o_agx_encode(a.b));
}
static inline RGBF o_agx_inverse(RGBF px) {
  RGBF a = c_splat(0.0f);
  a = c_add(a, c_scale(c3(1.19687900512017f, -0.0528968517574562f, -0.0529716355144438f), px.r));
  a = c_add(a, c_scale(c3(-0.0980208811401368f, 1.15190312990417f, -0.0980434501171241f), px.g));
  a = c_add(a, c_scale(c3(-0.0990297440797205f, -0.0989611768448433f, 1.15107367264116f), px.b));
  a = c3(fmaxf(a.r, 0.0f), fmaxf(a.g, 0.0f), fmaxf(a.b, 0.0f));
  return c3(o_srgb_to_linear(a.r), o_srgb_to_linear(a.g), o_srgb_to_linear(a.b));
}
static inline RGBF o_agx_look(RGBF px, float slope, float power, float saturation) {
  const float lum = c_luminance(px);
  px = c_mul(px, c_splat(slope));
  px = c3(o_pow(px.r, power), o_pow(px.g, power), o_pow(px.b, power));
  return c3(o_lerp(lum, px.r, saturation), o_lerp(lum, px.g, saturation), o_lerp(lum, px.b, saturation));
}
static inline RGBF o_tonemap_curve(const OracleOutputParams* p, RGBF px) {
  switch (p->tonemap) {
    case 1: return o_tonemap_aces(px);
    case 2: return o_tonemap_reinhard(px);
    case 3: return o_tonemap_uncharted2(px);
    case 4: return o_agx_inverse(o_agx_conversion(px));
    case 5: return o_agx_inverse(o_agx_look(o_agx_conversion(px), 1.0f, 1.35f, 1.4f));
    case 6: return o_agx_inverse(o_agx_look(o_agx_conversion(px), p->agx_slope, p->agx_power, p->agx_saturation));
    default: return px;
  }
}

/* purkinje.cuh:19-90 */
static inline RGBF o_purkinje_shift(const OracleOutputParams* p, RGBF px) {
  const float strength = 5000.0f;
  if (c_luminance(px) >= (1.0f / strength)) return px;
  const float lc = 0.096869562190332f * px.r + 0.318940374720484f * px.g - 0.188428411786113f * px.b;
  const float mc = 0.020208210904239f * px.r + 0.291385283197581f * px.g - 0.090918262127325f * px.b;
  const float sc = 0.002760510899553f * px.r - 0.008341563564118f * px.g + 0.067213551661950f * px.b;
  const float rod = -0.007607045462440f * px.r + 0.122492925567539f * px.g + 0.022445835141881f * px.b;
  const float k1 = p->purkinje_kappa1, k2 = p->purkinje_kappa2;
  const float lm = 1.0f / 0.63721f, mm = 1.0f / 0.39242f, sm = 1.0f / 1.6064f;
  const float ir = fmaxf(1.0f + (1.0f / 3.0f) * lm * (lc + k1 * rod), O_EPS);
  const float ig = fmaxf(1.0f + (1.0f / 3.0f) * mm * (mc + k1 * rod), O_EPS);
  const float ib = fmaxf(1.0f + (1.0f / 3.0f) * sm * (sc + k2 * rod), O_EPS);
  const float sr = o_rsqrt(ir), sg = o_rsqrt(ig), sb = o_rsqrt(ib);
  const float K = 45.0f, S = 10.0f, k3 = 0.6f, rw = 0.139f, pp = 0.6189f;
  RGBF opp = c3(((-k3 - rw) * sr + (1.0f + k3 * rw) * sg) * k1 * lm, (pp * k3 * sr + (1.0f - pp) * k3 * sg + sb) * k1 * mm,
                (pp * S * sr + (1.0f - pp) * S * sg) * k2 * sm);
  opp = c_scale(opp, (K / S) * rod);
  const float L = lc + 0.5f * (opp.b - opp.r), M = mc + 0.5f * (opp.b + opp.r), Sh = sc + opp.g + opp.b;
  const float X = 1.9102f * L - 1.1121f * M + 0.2019f * Sh, Y = 0.3710f * L + 0.6291f * M + 0.0000f * Sh,
              Z = 0.0000f * L + 0.0000f * M + 1.0000f * Sh;
  const RGBF srgb = c3(3.2405f * X - 1.5371f * Y - 0.4985f * Z, -0.9693f * X + 1.876f * Y + 0.0416f * Z, 0.0556f * X - 0.2040f * Y + 1.0572f * Z);
  float blend = o_saturate(1.0f - strength * c_luminance(px));
  blend = blend * blend;
  return c_add(c_scale(px, 1.0f - blend), c_scale(srgb, blend));
}

/* math.cuh:1483-1543 */
static inline RGBF o_rgb_to_hsv(RGBF c) {
  const float mx = fmaxf(c.r, fmaxf(c.g, c.b)), mn = fminf(c.r, fminf(c.g, c.b));
  const float s = (mx - mn) / mx;
  float h = 0.0f;
  if (s != 0.0f) {
    const float delta = mx - mn;
    if (mx == c.r) h = (c.g - c.b) / delta;
    else if (mx == c.g) h = 2.0f + (c.b - c.r) / delta;
    else h = 4.0f + (c.r - c.g) / delta;
    h = h * (1.0f / 6.0f);
    if (h < 0.0f) h = h + 1.0f;
  }
  return c3(h, s, mx);
}
static inline float o_hue_lobe(float h) { /* fmodf(h, 6) for h in [0, 12) */
  if (h >= 6.0f) h = h - 6.0f;
  return o_saturate(fabsf(h - 3.0f) - 1.0f);
}
static inline RGBF o_hsv_to_rgb(RGBF hsv) {
  const float s = hsv.g, v = hsv.b;
  if (s == 0.0f) return c_splat(v);
  const float h = hsv.r * 6.0f;
  const RGBF hue = c3(o_hue_lobe(h + 0.0f), o_hue_lobe(h + 4.0f), o_hue_lobe(h + 2.0f));
  return c_scale(c_add(c_scale(c_splat(1.0f), 1.0f - s), c_scale(hue, s)), v);
}

static inline float o_unit16(uint32_t v16) { return u2f(0x3F800000u | (v16 << 7)) - 1.0f; } /* random.cuh:150-154 */
static inline uint32_t o_squares16(uint32_t key, uint32_t counter) {                         /* random.cuh:197-212 */
  uint32_t x = counter * key, y = counter * key, z = y + key;
  x = x * x + y; x = swap16(x);
  x = x * x + z; x = swap16(x);
  return (x * x + y) >> 16;
}

/* tonemap.cuh:205-246 */
static inline RGBF o_display_transform(const OracleOutputParams* p, RGBF px, uint32_t x, uint32_t y) {
  if (p->passthrough) return px;
  if (p->purkinje) px = o_purkinje_shift(p, px);
  if (p->use_color_correction) {
    RGBF hsv = o_rgb_to_hsv(px);
    hsv = c_add(hsv, c3(p->cc_h, p->cc_s, p->cc_v));
    if (hsv.r < 0.0f) hsv.r = hsv.r + 1.0f;
    if (hsv.r > 1.0f) hsv.r = hsv.r - 1.0f;
    hsv.g = o_saturate(hsv.g);
    if (hsv.b < 0.0f) hsv.b = 0.0f;
    px = o_hsv_to_rgb(hsv);
  }
  px = c_scale(px, p->exposure);
  const float grain = p->film_grain * (o_unit16(o_squares16(0xfcbd6e15u, x + y * p->src_width)) - 0.5f);
  px = c3(fmaxf(0.0f, px.r + grain), fmaxf(0.0f, px.g + grain), fmaxf(0.0f, px.b + grain));
  return o_tonemap_curve(p, px);
}

/* post_common.cuh:6-59: width/height are the nominal output size, mem_scale = 2^-k addresses a coarser image in memory; the index
 * arithmetic is the reference's (uint * uint, then float) */
/* `last`: index of the plane's last element; the reference reads past the coarse image when the frame is not a multiple of the coarse block
 * (stale memory) - such indices are clamped so that the bytes stay a function of the input */
static inline uint32_t o_min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }
static inline float o_sample_plane(const float* plane, float x, float y, uint32_t width, uint32_t height, float mem_scale, uint32_t last) {
  x = fminf(fmaxf(x, 0.0f), u2f(0x3F7FFFFFu));
  y = fminf(fmaxf(y, 0.0f), u2f(0x3F7FFFFFu));
  const float sx = fmaxf(0.0f, x * (width - 1)) * mem_scale, sy = fmaxf(0.0f, y * (height - 1)) * mem_scale;
  const uint32_t x0 = (uint32_t) sx, y0 = (uint32_t) sy;
  uint32_t x1 = (uint32_t) (sx + mem_scale), y1 = (uint32_t) (sy + mem_scale);
  if (x1 > width - 1) x1 = width - 1;
  if (y1 > height - 1) y1 = height - 1;
  const uint32_t i00 = (uint32_t) ((float) x0 + (float) (y0 * width) * mem_scale), i01 = (uint32_t) ((float) x0 + (float) (y1 * width) * mem_scale);
  const uint32_t i10 = (uint32_t) ((float) x1 + (float) (y0 * width) * mem_scale), i11 = (uint32_t) ((float) x1 + (float) (y1 * width) * mem_scale);
  const float p00 = plane[o_min_u32(i00, last)], p01 = plane[o_min_u32(i01, last)], p10 = plane[o_min_u32(i10, last)], p11 = plane[o_min_u32(i11, last)];
  const float fx = sx - x0, ifx = 1.0f - fx, fy = sy - y0, ify = 1.0f - fy;
  float r = p00 * (ifx * ify);
  r += p01 * (ifx * fy);
  r += p10 * (fx * ify);
  r += p11 * (fx * fy);
  return r;
}
/* ---- bloom: device/device_post.c:10-170, cuda/post_common.cuh:46-149 ---- */
static inline float o_sample_plane_border(const float* plane, float x, float y, uint32_t width, uint32_t height, float weight) {
  if (x > u2f(0x3F7FFFFFu) || x < 0.0f) return 0.0f;
  if (y > u2f(0x3F7FFFFFu) || y < 0.0f) return 0.0f;
  const float sx = fmaxf(0.0f, x * (width - 1)), sy = fmaxf(0.0f, y * (height - 1)); /* fmaxf(0, NaN) = 0: 1-pixel levels */
  const uint32_t x0 = (uint32_t) sx, y0 = (uint32_t) sy;
  uint32_t x1 = (uint32_t) (sx + 1.0f), y1 = (uint32_t) (sy + 1.0f);
  if (x1 > width - 1) x1 = width - 1;
  if (y1 > height - 1) y1 = height - 1;
  const uint32_t i00 = (uint32_t) ((float) x0 + (float) (y0 * width)), i01 = (uint32_t) ((float) x0 + (float) (y1 * width));
  const uint32_t i10 = (uint32_t) ((float) x1 + (float) (y0 * width)), i11 = (uint32_t) ((float) x1 + (float) (y1 * width));
  const float p00 = plane[i00], p01 = plane[i01], p10 = plane[i10], p11 = plane[i11];
  const float fx = sx - x0, ifx = 1.0f - fx, fy = sy - y0, ify = 1.0f - fy;
  float r = p00 * (ifx * ify);
  r += p01 * (ifx * fy);
  r += p10 * (fx * ify);
  r += p11 * (fx * fy);
  return r * weight;
}
static void o_post_downsample(const float* src, uint32_t sw, uint32_t sh, float* dst, uint32_t tw, uint32_t th) {
  const float scale_x = 1.0f / (tw - 1), scale_y = 1.0f / (th - 1), step_x = 1.0f / (sw - 1), step_y = 1.0f / (sh - 1);
  for (uint32_t i = 0; i < tw * th; i++) {
    const uint32_t y = i / tw, x = i - y * tw;
    const float sx = scale_x * x, sy = scale_y * y;
    float p = 0.0f;
    p += o_sample_plane_border(src, sx - 0.5f * step_x, sy - 0.5f * step_y, sw, sh, 1.0f);
    p += o_sample_plane_border(src, sx + 0.5f * step_x, sy - 0.5f * step_y, sw, sh, 1.0f);
    p += o_sample_plane_border(src, sx - 0.5f * step_x, sy + 0.5f * step_y, sw, sh, 1.0f);
    p += o_sample_plane_border(src, sx + 0.5f * step_x, sy + 0.5f * step_y, sw, sh, 1.0f);
    p += o_sample_plane_border(src, sx, sy, sw, sh, 1.0f);
    p += o_sample_plane_border(src, sx, sy - step_y, sw, sh, 0.5f);
    p += o_sample_plane_border(src, sx - step_x, sy, sw, sh, 0.5f);
    p += o_sample_plane_border(src, sx + step_x, sy, sw, sh, 0.5f);
    p += o_sample_plane_border(src, sx, sy + step_y, sw, sh, 0.5f);
    p += o_sample_plane_border(src, sx - step_x, sy - step_y, sw, sh, 0.25f);
    p += o_sample_plane_border(src, sx + step_x, sy - step_y, sw, sh, 0.25f);
    p += o_sample_plane_border(src, sx - step_x, sy + step_y, sw, sh, 0.25f);
    p += o_sample_plane_border(src, sx + step_x, sy + step_y, sw, sh, 0.25f);
    p *= 1.0f / 8.0f;
    dst[i] = fmaxf(p, 0.0f);
  }
}
static void o_post_upsample(const float* src, uint32_t sw, uint32_t sh, float* dst, uint32_t tw, uint32_t th, float sa, float sb) {
  const float scale_x = 1.0f / (tw - 1), scale_y = 1.0f / (th - 1), step_x = 1.0f / (sw - 1), step_y = 1.0f / (sh - 1);
  for (uint32_t i = 0; i < tw * th; i++) {
    const uint32_t y = i / tw, x = i - y * tw;
    const float sx = scale_x * x, sy = scale_y * y;
    float p = o_sample_plane_border(src, sx - step_x, sy - step_y, sw, sh, 1.0f);
    p += o_sample_plane_border(src, sx, sy - step_y, sw, sh, 2.0f);
    p += o_sample_plane_border(src, sx + step_x, sy - step_y, sw, sh, 1.0f);
    p += o_sample_plane_border(src, sx - step_x, sy, sw, sh, 2.0f);
    p += o_sample_plane_border(src, sx, sy, sw, sh, 4.0f);
    p += o_sample_plane_border(src, sx + step_x, sy, sw, sh, 2.0f);
    p += o_sample_plane_border(src, sx - step_x, sy + step_y, sw, sh, 1.0f);
    p += o_sample_plane_border(src, sx, sy + step_y, sw, sh, 2.0f);
    p += o_sample_plane_border(src, sx + step_x, sy + step_y, sw, sh, 1.0f);
    p *= 1.0f / 20.0f;
    p *= sa;
    float base = dst[i];
    base *= sb;
    dst[i] = p + base;
  }
}
/* _device_post_bloom_apply: in place on the planar image of (full_width >> stage) x (full_height >> stage) */
static void output_bloom(float* image, uint32_t full_width, uint32_t full_height, uint32_t stage, float blend) {
  uint32_t chain = 0;
  for (uint32_t m = full_width < full_height ? full_width : full_height; m > 1; m >>= 1) chain++;
  if (stage + 1 >= chain) return;
  const uint32_t width = full_width >> stage, height = full_height >> stage, mips = chain - stage;
  float** mip = (float**) malloc(sizeof(float*) * mips);
  for (uint32_t i = 0; i < mips; i++) mip[i] = (float*) malloc(sizeof(float) * ((size_t) (width >> (i + 1)) * (height >> (i + 1)) + 1));
  for (uint32_t c = 0; c < 3; c++) {
    float* plane = image + (size_t) c * width * height;
    o_post_downsample(plane, width, height, mip[0], width >> 1, height >> 1);
    for (uint32_t i = 0; i + 1 < mips; i++) o_post_downsample(mip[i], width >> (i + 1), height >> (i + 1), mip[i + 1], width >> (i + 2), height >> (i + 2));
    for (uint32_t i = mips - 1; i > 0; i--) o_post_upsample(mip[i], width >> (i + 1), height >> (i + 1), mip[i - 1], width >> i, height >> i, 1.0f, 1.0f);
    o_post_upsample(mip[0], width >> 1, height >> 1, plane, width, height, blend / mips, 1.0f - blend);
  }
  for (uint32_t i = 0; i < mips; i++) free(mip[i]);
  free(mip);
}

static inline float o_dither_mask(const uint16_t* bn, uint32_t x, uint32_t y) { return o_unit16(bn[(x & 255u) + (y & 255u) * 256u]); }

/* math.cuh:1081-1168 */
static inline RGBF o_apply_filter(const OracleOutputParams* p, const uint16_t* bn, RGBF px, uint32_t x, uint32_t y) {
  switch (p->filter) {
    case 1: { const float v = c_luminance(px); return c_splat(v); }
    case 2: return c3(px.r * 0.393f + px.g * 0.769f + px.b * 0.189f, px.r * 0.349f + px.g * 0.686f + px.b * 0.168f, px.r * 0.272f + px.g * 0.534f + px.b * 0.131f);
    case 3: {
      const int tone = (int) (4.0f * c_luminance(px) + o_dither_mask(bn, x, y));
      if (tone == 0) return c3(15.0f / 255.0f, 56.0f / 255.0f, 15.0f / 255.0f);
      if (tone == 1) return c3(48.0f / 255.0f, 98.0f / 255.0f, 48.0f / 255.0f);
      if (tone == 2) return c3(139.0f / 255.0f, 172.0f / 255.0f, 15.0f / 255.0f);
      return c3(155.0f / 255.0f, 188.0f / 255.0f, 15.0f / 255.0f);
    }
    case 4: {
      const int tone = (int) (4.0f * c_luminance(px) + o_dither_mask(bn, x, y));
      if (tone == 0) return c_splat(0.0f);
      if (tone == 1) return c_splat(1.0f / 3.0f);
      if (tone == 2) return c_splat(2.0f / 3.0f);
      return c_splat(1.0f);
    }
    case 5: {
      px = c_scale(px, 1.5f);
      const uint32_t row = y % 3u;
      if (row == 0) { px.r = 0.0f; px.g = 0.0f; }
      else if (row == 1) { px.g = 0.0f; px.b = 0.0f; }
      else { px.r = 0.0f; px.b = 0.0f; }
      return px;
    }
    case 6: {
      const int tone = (int) (2.0f * c_luminance(px) + o_dither_mask(bn, x, y));
      return (tone == 0) ? c_splat(0.0f) : c_splat(1.0f);
    }
    default: return px;
  }
}

/* accumulation_generate_result_undersampling, accumulation.cuh:192-254: compact (width >> stage) x (height >> stage) planar image; block
 * (x, y) of 2^stage pixels shows the mean of the 4 - iteration pixels of it rendered so far */
static void output_result_undersampled(const float* first_moment, uint32_t width, uint32_t height, uint32_t stage, uint32_t iteration, float* result) {
  const uint32_t scale = 1u << stage, w = width >> stage, h = height >> stage, n = w * h;
  const size_t frame = (size_t) width * height;
  const float color_scale = 1.0f / (4 - iteration);
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t dst_y = i / w, dst_x = i - dst_y * w;
    const uint32_t base_x = dst_x << stage, base_y = dst_y << stage;
    RGBF sum = c_splat(0.0f);
    for (uint32_t id = iteration; id < 4; id++) {
      uint32_t px = base_x + ((id & 1u) ? 0u : scale >> 1), py = base_y + ((id & 2u) ? 0u : scale >> 1);
      if (px > width - 1) px = width - 1;
      if (py > height - 1) py = height - 1;
      const size_t index = px + (size_t) py * width;
      sum = c_add(sum, c3(first_moment[index], first_moment[frame + index], first_moment[2 * frame + index]));
    }
    sum = c_scale(sum, color_scale);
    result[i] = sum.r; result[n + i] = sum.g; result[2 * (size_t) n + i] = sum.b;
  }
}

/* generate_final_image (kernels.cuh:503-556) + convert_RGBF_to_ARGB8 (:558-644).
 * input: 3 planes of (src >> stage) pixels; frame_output: 3 planes of (src >> max(stage, supersampling)) pixels (display-referred RGB);
 * argb8: dst_width*dst_height words (b | g<<8 | r<<16 | a<<24) */
static void output_generate(const OracleOutputParams* p, const float* input, const uint16_t* bn, float* frame_output, uint32_t* argb8) {
  const uint32_t ui = p->undersampling_stage, uo = ui > p->supersampling ? ui : p->supersampling;
  const uint32_t output_scale = 1u << (uo - ui);
  const uint32_t out_w = p->src_width >> uo, out_h = p->src_height >> uo, in_w = p->src_width >> ui, in_h = p->src_height >> ui;
  const uint32_t ns = out_w * out_h;
  const size_t n_in = (size_t) in_w * in_h;
  const float norm = 1.0f / (output_scale * output_scale);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t) ns; i++) {
    const uint32_t y = (uint32_t) i / out_w, x = (uint32_t) i - y * out_w;
    const uint32_t source_x = x * output_scale, source_y = y * output_scale;
    RGBF color = c_splat(0.0f);
    for (uint32_t yi = 0; yi < output_scale; yi++) {
      for (uint32_t xi = 0; xi < output_scale; xi++) {
        uint32_t px_x = source_x + xi, px_y = source_y + yi;
        if (px_x > in_w - 1) px_x = in_w - 1;
        if (px_y > in_h - 1) px_y = in_h - 1;
        const size_t index = px_x + (size_t) px_y * in_w;
        const RGBF px = c3(input[index] * p->inv_sample_count, input[n_in + index] * p->inv_sample_count, input[2 * n_in + index] * p->inv_sample_count);
        color = c_add(color, o_display_transform(p, px, px_x, px_y));
      }
    }
    color = c_scale(color, norm);
    frame_output[i] = color.r; frame_output[ns + i] = color.g; frame_output[2 * (size_t) ns + i] = color.b;
  }
  const uint32_t um = uo - p->supersampling;
  const uint32_t nominal_w = p->src_width >> p->supersampling, nominal_h = p->src_height >> p->supersampling;
  const uint32_t n = p->dst_width * p->dst_height;
  const float scale_x = 1.0f / (p->dst_width - 1), scale_y = 1.0f / (p->dst_height - 1);
  const float mem_scale = 1.0f / (1u << um);
  const bool scaled = p->dst_width != nominal_w || p->dst_height != nominal_h;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t) n; i++) {
    const uint32_t y = (uint32_t) i / p->dst_width, x = (uint32_t) i - y * p->dst_width;
    RGBF px;
    if (scaled) {
      const float sx = x * scale_x, sy = y * scale_y;
      px = c3(o_sample_plane(frame_output, sx, sy, nominal_w, nominal_h, mem_scale, ns - 1), o_sample_plane(frame_output + ns, sx, sy, nominal_w, nominal_h, mem_scale, ns - 1),
              o_sample_plane(frame_output + 2 * (size_t) ns, sx, sy, nominal_w, nominal_h, mem_scale, ns - 1));
    }
    else {
      const size_t src = o_min_u32(x >> um, out_w - 1) + (size_t) o_min_u32(y >> um, out_h - 1) * out_w; /* the edge repeats where the reference reads past the image */
      px = c3(frame_output[src], frame_output[ns + src], frame_output[2 * (size_t) ns + src]);
    }
    px = o_apply_filter(p, bn, px, x, y);
    const float dither = p->dithering ? o_dither_mask(bn, x, y) : 0.5f;
    const float r = fmaxf(0.0f, fminf(255.9999f, dither + 255.0f * o_linear_to_srgb(px.r)));
    const float g = fmaxf(0.0f, fminf(255.9999f, dither + 255.0f * o_linear_to_srgb(px.g)));
    const float b = fmaxf(0.0f, fminf(255.9999f, dither + 255.0f * o_linear_to_srgb(px.b)));
    argb8[i] = 0xFF000000u | (f2u_sat(r) << 16) | (f2u_sat(g) << 8) | f2u_sat(b);
  }
}

#endif
