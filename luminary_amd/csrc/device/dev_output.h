// Output chain: accumulated moments -> display-referred RGB -> ARGB8.
// Reference: cuda/accumulation.cuh:86-153 (result = first moment / sample count), cuda/kernels.cuh:503-556 (generate_final_image),
// cuda/tonemap.cuh (exposure, colour correction, film grain, Purkinje shift, tone curves), cuda/kernels.cuh:558-644
// (convert_RGBF_to_ARGB8: optional bilinear resize, filters, dither, sRGB), cuda/math.cuh:1040-1170, :1483-1543, cuda/purkinje.cuh,
// cuda/post_common.cuh:6-50, cuda/random.cuh:144-154, :197-212, :370-379.
// Scope of this implementation: any supersampling and undersampling stage, bloom.
// Numerics: the reference uses fast-math log2f/powf/rsqrtf whose bits are unspecified; here log2, exp2 and pow are the fixed
// sequences below (relative error < 3e-7), mirrored operation by operation in oracle/o_output.h, so the bytes of an image are a
// pure function of the moments on any device.
#pragma once

#include "dev_sampler.h"

LUM_NS_BEGIN

struct OutputParams {
  uint32_t src_width, src_height;  // rendered frame
  uint32_t dst_width, dst_height;  // requested image
  float inv_sample_count, exposure;
  uint32_t tonemap, filter, dithering, purkinje, use_color_correction, passthrough;  // passthrough: shading mode != default
  float purkinje_kappa1, purkinje_kappa2;
  float cc_h, cc_s, cc_v;
  float film_grain;
  float agx_slope, agx_power, agx_saturation;
  // settings.supersampling (the rendered frame is the nominal output size << supersampling) and the stage of the undersampling preview
  // the input image belongs to (0 = a full result image; s > 0 = the compact (src >> s) image of k_result_undersampled)
  uint32_t supersampling, undersampling_stage;
};

LUM_DEV float linear_to_srgb(float v) { return (v <= 0.0031308f) ? 12.92f * v : 1.055f * pow_det(v, 0.416666666667f) - 0.055f; }  // math.cuh:1044-1051
LUM_DEV float srgb_to_linear(float v) { return (v <= 0.04045f) ? v / 12.92f : pow_det((v + 0.055f) / 1.055f, 2.4f); }             // math.cuh:1053-1060

// ---- tone curves (tonemap.cuh) ----
LUM_DEV Col tonemap_aces(Col px) {
  Col c = col(0.59719f * px.r + 0.35458f * px.g + 0.04823f * px.b, 0.07600f * px.r + 0.90834f * px.g + 0.01566f * px.b,
              0.02840f * px.r + 0.13383f * px.g + 0.83777f * px.b);
  Col a = c + splat(0.0245786f);
  a = c * a;
  a = a + splat(-0.000090537f);
  Col b = c * splat(0.983729f);
  b = b + splat(0.432951f);
  b = c * b;
  b = b + splat(0.238081f);
  b = col(1.0f / b.r, 1.0f / b.g, 1.0f / b.b);
  c = a * b;
  return col(1.60475f * c.r - 0.53108f * c.g - 0.07367f * c.b, -0.10208f * c.r + 1.10813f * c.g - 0.00605f * c.b,
             -0.00327f * c.r - 0.07276f * c.g + 1.07602f * c.b);
}
LUM_DEV float uncharted2_partial(float v) {
  const float a = 0.15f, b = 0.50f, c = 0.10f, d = 0.20f, e = 0.02f, f = 0.30f;
  return ((v * (a * v + c * b) + d * e) / (v * (a * v + b) + d * f)) - e / f;
}
LUM_DEV Col tonemap_uncharted2(Col px) {
  px = px * splat(2.0f);
  px = col(uncharted2_partial(px.r), uncharted2_partial(px.g), uncharted2_partial(px.b));
  const float s = 1.0f / uncharted2_partial(11.2f);
  return px * splat(s);
}
LUM_DEV Col tonemap_reinhard(Col px) { const float f = 1.0f / (1.0f + luminance(px)); return col(px.r * f, px.g * f, px.b * f); }
LUM_DEV float agx_contrast(float v) {
  const float v2 = v * v, v4 = v2 * v2;
  return 15.5f * v4 * v2 - 40.14f * v4 * v + 31.96f * v4 - 6.868f * v2 * v + 0.4298f * v2 + 0.1191f * v - 0.00232f;
}
LUM_DEV float agx_encode(float v) {
  const float lo = -12.47393f, hi = 4.026069f;
  v = fmaxf(v, 0.00017578139f);
  v = fminf(fmaxf(log2_det(v), lo), hi);
  return agx_contrast((v - lo) / (hi - lo));
}
LUM_DEV Col agx_conversion(Col px) {
  Col a = splat(0.0f);
  a = a + col(0.842479062253094f, 0.0423282422610123f, 0.0423756549057051f) * px.r;
  a = a + col(0.0784335999999992f, 0.878468636469772f, 0.0784336f) * px.g;
  a = a + col(0.0792237451477643f, 0.0791661274605434f, 0.879142973793104f) * px.b;
  return col(agx_encode(a.r), agx_encode(a.g), agx_encode(a.b));
}
LUM_DEV Col agx_inverse(Col px) {
  Col a = splat(0.0f);
  a = a + col(1.19687900512017f, -0.0528968517574562f, -0.0529716355144438f) * px.r;
  a = a + col(-0.0980208811401368f, 1.15190312990417f, -0.0980434501171241f) * px.g;
  a = a + col(-0.0990297440797205f, -0.0989611768448433f, 1.15107367264116f) * px.b;
  a = col(fmaxf(a.r, 0.0f), fmaxf(a.g, 0.0f), fmaxf(a.b, 0.0f));
  return col(srgb_to_linear(a.r), srgb_to_linear(a.g), srgb_to_linear(a.b));
}
LUM_DEV Col agx_look(Col px, float slope, float power, float saturation) {
  const float lum = luminance(px);
  px = px * splat(slope);
  px = col(pow_det(px.r, power), pow_det(px.g, power), pow_det(px.b, power));
  return col(lerpf(lum, px.r, saturation), lerpf(lum, px.g, saturation), lerpf(lum, px.b, saturation));
}
LUM_DEV Col tonemap_curve(const OutputParams& p, Col px) {
  switch (p.tonemap) {
    case 1: return tonemap_aces(px);
    case 2: return tonemap_reinhard(px);
    case 3: return tonemap_uncharted2(px);
    case 4: return agx_inverse(agx_conversion(px));
    case 5: return agx_inverse(agx_look(agx_conversion(px), 1.0f, 1.35f, 1.4f));
    case 6: return agx_inverse(agx_look(agx_conversion(px), p.agx_slope, p.agx_power, p.agx_saturation));
    default: return px;
  }
}

// ---- Purkinje shift (purkinje.cuh) ----
LUM_DEV Col purkinje_shift(const OutputParams& p, Col px) {
  const float strength = 5000.0f;
  if (luminance(px) >= (1.0f / strength)) return px;
  const float lc = 0.096869562190332f * px.r + 0.318940374720484f * px.g - 0.188428411786113f * px.b;
  const float mc = 0.020208210904239f * px.r + 0.291385283197581f * px.g - 0.090918262127325f * px.b;
  const float sc = 0.002760510899553f * px.r - 0.008341563564118f * px.g + 0.067213551661950f * px.b;
  const float rod = -0.007607045462440f * px.r + 0.122492925567539f * px.g + 0.022445835141881f * px.b;
  const float k1 = p.purkinje_kappa1, k2 = p.purkinje_kappa2;
  const float lm = 1.0f / 0.63721f, mm = 1.0f / 0.39242f, sm = 1.0f / 1.6064f;
  const float ir = fmaxf(1.0f + (1.0f / 3.0f) * lm * (lc + k1 * rod), kEps);
  const float ig = fmaxf(1.0f + (1.0f / 3.0f) * mm * (mc + k1 * rod), kEps);
  const float ib = fmaxf(1.0f + (1.0f / 3.0f) * sm * (sc + k2 * rod), kEps);
  const float sr = rsqrt_ieee(ir), sg = rsqrt_ieee(ig), sb = rsqrt_ieee(ib);
  const float K = 45.0f, S = 10.0f, k3 = 0.6f, rw = 0.139f, pp = 0.6189f;
  Col opp = col(((-k3 - rw) * sr + (1.0f + k3 * rw) * sg) * k1 * lm, (pp * k3 * sr + (1.0f - pp) * k3 * sg + sb) * k1 * mm,
                (pp * S * sr + (1.0f - pp) * S * sg) * k2 * sm);
  opp = opp * ((K / S) * rod);
  const float L = lc + 0.5f * (opp.b - opp.r), M = mc + 0.5f * (opp.b + opp.r), Sh = sc + opp.g + opp.b;
  const float X = 1.9102f * L - 1.1121f * M + 0.2019f * Sh, Y = 0.3710f * L + 0.6291f * M + 0.0000f * Sh,
              Z = 0.0000f * L + 0.0000f * M + 1.0000f * Sh;
  const Col srgb = col(3.2405f * X - 1.5371f * Y - 0.4985f * Z, -0.9693f * X + 1.876f * Y + 0.0416f * Z, 0.0556f * X - 0.2040f * Y + 1.0572f * Z);
  float blend = saturate(1.0f - strength * luminance(px));
  blend = blend * blend;
  return px * (1.0f - blend) + srgb * blend;
}

// ---- colour correction in HSV (math.cuh:1483-1543; fmodf of a value in [0, 12) by 6 is one conditional subtraction) ----
LUM_DEV Col rgb_to_hsv(Col c) {
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
  return col(h, s, mx);
}
LUM_DEV float hue_lobe(float h) {
  if (h >= 6.0f) h = h - 6.0f;
  return saturate(fabsf(h - 3.0f) - 1.0f);
}
LUM_DEV Col hsv_to_rgb(Col hsv) {
  const float s = hsv.g, v = hsv.b;
  if (s == 0.0f) return splat(v);
  const float h = hsv.r * 6.0f;
  const Col hue = col(hue_lobe(h + 0.0f), hue_lobe(h + 4.0f), hue_lobe(h + 2.0f));
  return (splat(1.0f) * (1.0f - s) + hue * s) * v;
}

// random.cuh:150-154, :197-212, :305-307
LUM_DEV float unit_float16(uint32_t v16) { return bitsf(0x3F800000u | (v16 << 7)) - 1.0f; }
LUM_DEV uint32_t squares16(uint32_t key, uint32_t counter) {
  uint32_t x = counter * key, y = counter * key, z = y + key;
  x = x * x + y; x = swap_halves(x);
  x = x * x + z; x = swap_halves(x);
  return (x * x + y) >> 16;
}

// tonemap_apply, tonemap.cuh:205-246
LUM_DEV Col display_transform(const OutputParams& p, Col px, uint32_t x, uint32_t y) {
  if (p.passthrough) return px;
  if (p.purkinje) px = purkinje_shift(p, px);
  if (p.use_color_correction) {
    Col hsv = rgb_to_hsv(px);
    hsv = hsv + col(p.cc_h, p.cc_s, p.cc_v);
    if (hsv.r < 0.0f) hsv.r = hsv.r + 1.0f;
    if (hsv.r > 1.0f) hsv.r = hsv.r - 1.0f;
    hsv.g = saturate(hsv.g);
    if (hsv.b < 0.0f) hsv.b = 0.0f;
    px = hsv_to_rgb(hsv);
  }
  px = px * p.exposure;
  const float grain = p.film_grain * (unit_float16(squares16(0xfcbd6e15u, x + y * p.src_width)) - 0.5f);
  px = col(fmaxf(0.0f, px.r + grain), fmaxf(0.0f, px.g + grain), fmaxf(0.0f, px.b + grain));
  return tonemap_curve(p, px);
}

#if !LUM_FAST  // flavour-neutral: compiled once, in the exact translation unit
// generate_final_image, kernels.cuh:503-556 (with accumulation_generate_result's division by the sample count folded in): planar input
// image of (src >> stage) pixels -> planar display-referred RGB of (src >> max(stage, supersampling)) pixels; every output pixel is the
// mean of the output_scale^2 tone-mapped input pixels below it, summed row by row.
__global__ __launch_bounds__(256) void k_final_image(OutputParams p, const float* __restrict__ input, float* __restrict__ frame_output) {
  const uint32_t ui = p.undersampling_stage, uo = max(ui, p.supersampling);
  const uint32_t output_scale = 1u << (uo - ui);
  const uint32_t out_w = p.src_width >> uo, out_h = p.src_height >> uo, in_w = p.src_width >> ui, in_h = p.src_height >> ui;
  const uint32_t n = out_w * out_h, n_in = in_w * in_h;
  const float norm = 1.0f / (output_scale * output_scale);
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t y = i / out_w, x = i - y * out_w;
    const uint32_t source_x = x * output_scale, source_y = y * output_scale;
    Col color = splat(0.0f);
    for (uint32_t yi = 0; yi < output_scale; yi++) {
      for (uint32_t xi = 0; xi < output_scale; xi++) {
        const uint32_t px_x = min(source_x + xi, in_w - 1), px_y = min(source_y + yi, in_h - 1);
        const uint32_t index = px_x + px_y * in_w;
        Col px = col(input[index] * p.inv_sample_count, input[n_in + index] * p.inv_sample_count, input[2 * n_in + index] * p.inv_sample_count);
        color = color + display_transform(p, px, px_x, px_y);
      }
    }
    color = color * norm;
    frame_output[i] = color.r; frame_output[n + i] = color.g; frame_output[2 * n + i] = color.b;
  }
}
#endif

#if !LUM_FAST  // flavour-neutral: compiled once, in the exact translation unit
// accumulation_generate_result_undersampling, accumulation.cuh:192-254: while the first sample is rendered coarse to fine, block (x, y) of
// 2^stage pixels shows the mean of the 4 - iteration pixels of it that exist so far (pattern of kernels.cuh:20-45). Output: compact
// planar image of (width >> stage) x (height >> stage).
__global__ __launch_bounds__(256) void k_result_undersampled(const float* __restrict__ first_moment, uint32_t width, uint32_t height, uint32_t stage, uint32_t iteration,
                                                             float* __restrict__ result) {
  const uint32_t scale = 1u << stage, w = width >> stage, h = height >> stage, n = w * h, frame = width * height;
  const float color_scale = 1.0f / (4 - iteration);
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t dst_y = i / w, dst_x = i - dst_y * w;
    const uint32_t base_x = dst_x << stage, base_y = dst_y << stage;
    Col sum = splat(0.0f);
    for (uint32_t id = iteration; id < 4; id++) {
      const uint32_t px = min(base_x + ((id & 1u) ? 0u : scale >> 1), width - 1), py = min(base_y + ((id & 2u) ? 0u : scale >> 1), height - 1);
      const uint32_t index = px + py * width;
      sum = sum + col(first_moment[index], first_moment[frame + index], first_moment[2 * frame + index]);
    }
    sum = sum * color_scale;
    result[i] = sum.r; result[n + i] = sum.g; result[2 * n + i] = sum.b;
  }
}
#endif

// post_sample_buffer_clamp, post_common.cuh:6-59. `width`/`height` are the nominal output size; a coarser image in memory is addressed
// through mem_scale = 2^-k, whose index arithmetic the reference carries out in float (kept: it decides the rounding of the row offset).
// `last`: index of the plane's last element. A frame that is not a multiple of the coarse block makes the reference read past the coarse
// image (stale memory); such indices are clamped here so that the bytes stay a function of the input.
LUM_DEV float sample_plane(const float* __restrict__ plane, float x, float y, uint32_t width, uint32_t height, float mem_scale, uint32_t last) {
  x = fminf(fmaxf(x, 0.0f), bitsf(0x3F7FFFFFu));
  y = fminf(fmaxf(y, 0.0f), bitsf(0x3F7FFFFFu));
  const float sx = fmaxf(0.0f, x * (width - 1)) * mem_scale, sy = fmaxf(0.0f, y * (height - 1)) * mem_scale;
  const uint32_t x0 = (uint32_t) sx, y0 = (uint32_t) sy;
  const uint32_t x1 = min((uint32_t) (sx + mem_scale), width - 1), y1 = min((uint32_t) (sy + mem_scale), height - 1);
  const uint32_t i00 = (uint32_t) ((float) x0 + (float) (y0 * width) * mem_scale), i01 = (uint32_t) ((float) x0 + (float) (y1 * width) * mem_scale);
  const uint32_t i10 = (uint32_t) ((float) x1 + (float) (y0 * width) * mem_scale), i11 = (uint32_t) ((float) x1 + (float) (y1 * width) * mem_scale);
  const float p00 = plane[min(i00, last)], p01 = plane[min(i01, last)], p10 = plane[min(i10, last)], p11 = plane[min(i11, last)];
  const float fx = sx - x0, ifx = 1.0f - fx, fy = sy - y0, ify = 1.0f - fy;
  float r = p00 * (ifx * ify);
  r += p01 * (ifx * fy);
  r += p10 * (fx * ify);
  r += p11 * (fx * fy);
  return r;
}

// ---- bloom (device/device_post.c:10-170, cuda/post_common.cuh:6-149): a mip chain of the result image, 13-tap downsampling, 9-tap tent
// upsampling, blended back into the image. Planes are processed one at a time. ----
// post_sample_buffer_border: zero outside [0, 1); no clamp inside. A NaN coordinate (a 1-pixel-wide level: 1 / (1 - 1) * 0) passes both
// comparisons and fmaxf(0, NaN) = 0 sends it to texel 0, as on the reference's hardware.
LUM_DEV float sample_plane_border(const float* __restrict__ plane, float x, float y, uint32_t width, uint32_t height, float weight) {
  if (x > bitsf(0x3F7FFFFFu) || x < 0.0f) return 0.0f;
  if (y > bitsf(0x3F7FFFFFu) || y < 0.0f) return 0.0f;
  const float sx = fmaxf(0.0f, x * (width - 1)), sy = fmaxf(0.0f, y * (height - 1));
  const uint32_t x0 = (uint32_t) sx, y0 = (uint32_t) sy;
  const uint32_t x1 = min((uint32_t) (sx + 1.0f), width - 1), y1 = min((uint32_t) (sy + 1.0f), height - 1);
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
#if !LUM_FAST  // flavour-neutral: compiled once, in the exact translation unit
__global__ __launch_bounds__(256) void k_post_downsample(const float* __restrict__ src, uint32_t sw, uint32_t sh, float* __restrict__ dst, uint32_t tw, uint32_t th) {
  const float scale_x = 1.0f / (tw - 1), scale_y = 1.0f / (th - 1), step_x = 1.0f / (sw - 1), step_y = 1.0f / (sh - 1);
  const uint32_t n = tw * th;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t y = i / tw, x = i - y * tw;
    const float sx = scale_x * x, sy = scale_y * y;
    float p = 0.0f;
    p += sample_plane_border(src, sx - 0.5f * step_x, sy - 0.5f * step_y, sw, sh, 1.0f);
    p += sample_plane_border(src, sx + 0.5f * step_x, sy - 0.5f * step_y, sw, sh, 1.0f);
    p += sample_plane_border(src, sx - 0.5f * step_x, sy + 0.5f * step_y, sw, sh, 1.0f);
    p += sample_plane_border(src, sx + 0.5f * step_x, sy + 0.5f * step_y, sw, sh, 1.0f);
    p += sample_plane_border(src, sx, sy, sw, sh, 1.0f);
    p += sample_plane_border(src, sx, sy - step_y, sw, sh, 0.5f);
    p += sample_plane_border(src, sx - step_x, sy, sw, sh, 0.5f);
    p += sample_plane_border(src, sx + step_x, sy, sw, sh, 0.5f);
    p += sample_plane_border(src, sx, sy + step_y, sw, sh, 0.5f);
    p += sample_plane_border(src, sx - step_x, sy - step_y, sw, sh, 0.25f);
    p += sample_plane_border(src, sx + step_x, sy - step_y, sw, sh, 0.25f);
    p += sample_plane_border(src, sx - step_x, sy + step_y, sw, sh, 0.25f);
    p += sample_plane_border(src, sx + step_x, sy + step_y, sw, sh, 0.25f);
    p *= 1.0f / 8.0f;
    dst[i] = fmaxf(p, 0.0f);  // threshold 0 (device_post.c:82)
  }
}
#endif
#if !LUM_FAST  // flavour-neutral: compiled once, in the exact translation unit
// dst may be the base image (every thread reads only its own base pixel)
__global__ __launch_bounds__(256) void k_post_upsample(const float* __restrict__ src, uint32_t sw, uint32_t sh, float* dst, uint32_t tw, uint32_t th, float sa, float sb) {
  const float scale_x = 1.0f / (tw - 1), scale_y = 1.0f / (th - 1), step_x = 1.0f / (sw - 1), step_y = 1.0f / (sh - 1);
  const uint32_t n = tw * th;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t y = i / tw, x = i - y * tw;
    const float sx = scale_x * x, sy = scale_y * y;
    float p = sample_plane_border(src, sx - step_x, sy - step_y, sw, sh, 1.0f);
    p += sample_plane_border(src, sx, sy - step_y, sw, sh, 2.0f);
    p += sample_plane_border(src, sx + step_x, sy - step_y, sw, sh, 1.0f);
    p += sample_plane_border(src, sx - step_x, sy, sw, sh, 2.0f);
    p += sample_plane_border(src, sx, sy, sw, sh, 4.0f);
    p += sample_plane_border(src, sx + step_x, sy, sw, sh, 2.0f);
    p += sample_plane_border(src, sx - step_x, sy + step_y, sw, sh, 1.0f);
    p += sample_plane_border(src, sx, sy + step_y, sw, sh, 2.0f);
    p += sample_plane_border(src, sx + step_x, sy + step_y, sw, sh, 1.0f);
    p *= 1.0f / 20.0f;
    p *= sa;
    float base = dst[i];
    base *= sb;
    dst[i] = p + base;
  }
}
#endif

LUM_DEV float dither_mask(const uint16_t* __restrict__ bluenoise_1d, uint32_t x, uint32_t y) { return unit_float16(bluenoise_1d[(x & 255u) + (y & 255u) * 256u]); }

// math.cuh:1081-1168
LUM_DEV Col apply_filter(const OutputParams& p, const uint16_t* __restrict__ bn, Col px, uint32_t x, uint32_t y) {
  switch (p.filter) {
    case 1: { const float v = luminance(px); return splat(v); }
    case 2: return col(px.r * 0.393f + px.g * 0.769f + px.b * 0.189f, px.r * 0.349f + px.g * 0.686f + px.b * 0.168f, px.r * 0.272f + px.g * 0.534f + px.b * 0.131f);
    case 3: {
      const int tone = (int) (4.0f * luminance(px) + dither_mask(bn, x, y));
      if (tone == 0) return col(15.0f / 255.0f, 56.0f / 255.0f, 15.0f / 255.0f);
      if (tone == 1) return col(48.0f / 255.0f, 98.0f / 255.0f, 48.0f / 255.0f);
      if (tone == 2) return col(139.0f / 255.0f, 172.0f / 255.0f, 15.0f / 255.0f);
      return col(155.0f / 255.0f, 188.0f / 255.0f, 15.0f / 255.0f);
    }
    case 4: {
      const int tone = (int) (4.0f * luminance(px) + dither_mask(bn, x, y));
      if (tone == 0) return splat(0.0f);
      if (tone == 1) return splat(1.0f / 3.0f);
      if (tone == 2) return splat(2.0f / 3.0f);
      return splat(1.0f);
    }
    case 5: {
      px = px * 1.5f;
      const uint32_t row = y % 3u;
      if (row == 0) { px.r = 0.0f; px.g = 0.0f; }
      else if (row == 1) { px.g = 0.0f; px.b = 0.0f; }
      else { px.r = 0.0f; px.b = 0.0f; }
      return px;
    }
    case 6: {
      const int tone = (int) (2.0f * luminance(px) + dither_mask(bn, x, y));
      return (tone == 0) ? splat(0.0f) : splat(1.0f);
    }
    default: return px;
  }
}

#if !LUM_FAST  // flavour-neutral: compiled once, in the exact translation unit
// convert_RGBF_to_ARGB8, kernels.cuh:558-644 (bytes b, g, r, a)
__global__ __launch_bounds__(256) void k_to_argb8(OutputParams p, const float* __restrict__ frame_output, const uint16_t* __restrict__ bluenoise_1d,
                                                  uint32_t* __restrict__ dst) {
  const uint32_t uo = max(p.undersampling_stage, p.supersampling), um = uo - p.supersampling;
  const uint32_t nominal_w = p.src_width >> p.supersampling, nominal_h = p.src_height >> p.supersampling;  // the size the frame is rendered for
  const uint32_t mem_w = p.src_width >> uo, mem_h = p.src_height >> uo, ns = mem_w * mem_h;                // the image in memory
  const uint32_t n = p.dst_width * p.dst_height;
  const float scale_x = 1.0f / (p.dst_width - 1), scale_y = 1.0f / (p.dst_height - 1);
  const float mem_scale = 1.0f / (1u << um);
  const bool scaled = p.dst_width != nominal_w || p.dst_height != nominal_h;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t y = i / p.dst_width, x = i - y * p.dst_width;
    Col px;
    if (scaled) {
      const float sx = x * scale_x, sy = y * scale_y;
      px = col(sample_plane(frame_output, sx, sy, nominal_w, nominal_h, mem_scale, ns - 1), sample_plane(frame_output + ns, sx, sy, nominal_w, nominal_h, mem_scale, ns - 1),
               sample_plane(frame_output + 2 * ns, sx, sy, nominal_w, nominal_h, mem_scale, ns - 1));
    }
    else {
      const uint32_t src = min(x >> um, mem_w - 1) + min(y >> um, mem_h - 1) * mem_w;  // the edge repeats where the reference reads past the coarse image
      px = col(frame_output[src], frame_output[ns + src], frame_output[2 * ns + src]);
    }
    px = apply_filter(p, bluenoise_1d, px, x, y);
    const float dither = p.dithering ? dither_mask(bluenoise_1d, x, y) : 0.5f;
    const float r = fmaxf(0.0f, fminf(255.9999f, dither + 255.0f * linear_to_srgb(px.r)));
    const float g = fmaxf(0.0f, fminf(255.9999f, dither + 255.0f * linear_to_srgb(px.g)));
    const float b = fmaxf(0.0f, fminf(255.9999f, dither + 255.0f * linear_to_srgb(px.b)));
    dst[i] = 0xFF000000u | (f2u_sat(r) << 16) | (f2u_sat(g) << 8) | f2u_sat(b);
  }
}
#endif

LUM_NS_END
