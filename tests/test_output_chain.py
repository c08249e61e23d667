"""Output chain (tone mapping, filters, dither, ARGB8): oracle self-checks on the CPU and HIP == oracle on the GPU."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from luminary_amd.core import OutputParams, default_output_params


def _synthetic_moment(w, h, spp, seed=0):
    """Planar first moment of a frame with dark, mid and very bright regions (values are sums over `spp` samples)."""
    rng = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    base = np.exp2(-14.0 + 22.0 * (x / max(w - 1, 1)))  # 6e-5 .. 256
    img = np.stack([base * (0.6 + 0.4 * np.sin(0.3 * y)), base * (0.5 + 0.5 * rng.rand(h, w)), base * (0.2 + 0.8 * y / max(h - 1, 1))]).astype(np.float32)
    img[:, : h // 8, : w // 8] = 0.0
    return (img * spp).reshape(3, -1).astype(np.float32)


def _unpack(argb):
    return np.stack([(argb >> 16) & 0xFF, (argb >> 8) & 0xFF, argb & 0xFF, argb >> 24], axis=-1).astype(np.int32)


def test_log2_exp2_pow_accuracy():
    l = oracle_lib.lib()
    for f in ("oracle_log2", "oracle_exp2"):
        getattr(l, f).restype = C.c_float
        getattr(l, f).argtypes = [C.c_float]
    l.oracle_pow.restype = C.c_float
    l.oracle_pow.argtypes = [C.c_float, C.c_float]
    xs = np.exp(np.random.RandomState(0).uniform(-20, 20, 4000)).astype(np.float32)
    got = np.array([l.oracle_log2(float(x)) for x in xs])
    assert np.abs(got - np.log2(xs.astype(np.float64))).max() < 3e-6
    ys = np.random.RandomState(1).uniform(-100, 100, 4000).astype(np.float32)
    got = np.array([l.oracle_exp2(float(v)) for v in ys])
    assert (np.abs(got - np.exp2(ys.astype(np.float64))) / np.exp2(ys.astype(np.float64))).max() < 3e-7
    assert l.oracle_exp2(3.0) == 8.0 and l.oracle_exp2(-1.0) == 0.5 and l.oracle_log2(8.0) == 3.0 and l.oracle_pow(0.0, 2.4) == 0.0
    v = np.random.RandomState(2).uniform(0.004, 1.0, 4000).astype(np.float32)
    got = np.array([l.oracle_pow(float(a), 2.4) for a in v])
    assert (np.abs(got - v.astype(np.float64) ** 2.4) / got).max() < 4e-6


def test_oracle_output_properties():
    w, h, spp = 96, 40, 4
    fm = _synthetic_moment(w, h, spp)
    p = default_output_params(w, h, spp)
    p.dithering = 0
    argb, planes = oracle_lib.generate_output(p, fm)
    px = _unpack(argb)
    assert (px[..., 3] == 255).all() and (px[: h // 8, : w // 8, :3] == 0).all()  # opaque alpha, black stays black
    assert px[..., :3].max() >= 250 and np.isfinite(planes).all() and planes.min() >= 0.0
    # no tone curve, no Purkinje shift: the bytes are the sRGB transfer function of the mean radiance
    p.tonemap, p.purkinje = 0, 0
    argb, planes = oracle_lib.generate_output(p, fm)
    mean = (fm / spp).reshape(3, h, w).astype(np.float64)
    assert np.allclose(planes, mean, rtol=1e-6, atol=0)
    srgb = np.where(mean <= 0.0031308, 12.92 * mean, 1.055 * np.power(np.maximum(mean, 1e-30), 1 / 2.4) - 0.055)
    want = np.clip(np.floor(0.5 + 255.0 * srgb), 0, 255)
    got = _unpack(argb)[..., :3].transpose(2, 0, 1)
    assert np.abs(got - want).max() <= 1 and (got != want).mean() < 1e-3
    # dithering only moves a value by at most one code
    p.dithering = 1
    d = _unpack(oracle_lib.generate_output(p, fm)[0])[..., :3].transpose(2, 0, 1)
    assert np.abs(d - got).max() <= 1 and (d != got).any()
    # exposure is a multiplication before the curve; filters act after it
    p.dithering, p.exposure = 0, 2.0
    assert np.allclose(oracle_lib.generate_output(p, fm)[1], 2.0 * planes, rtol=1e-6)
    p.exposure, p.filter = 1.0, 1
    g = _unpack(oracle_lib.generate_output(p, fm)[0])
    assert (g[..., 0] == g[..., 1]).all() and (g[..., 1] == g[..., 2]).all()
    p.filter = 6
    bw = _unpack(oracle_lib.generate_output(p, fm)[0])[..., :3]
    assert set(np.unique(bw)) <= {0, 255}
    # every tone curve is monotone in a grey ramp and maps black to (almost) black
    ramp = np.tile(np.exp2(np.linspace(-12, 6, w, dtype=np.float32)), (3, h, 1)).reshape(3, -1)
    for tm in range(7):
        q = default_output_params(w, h, 1)
        q.tonemap, q.purkinje, q.dithering = tm, 0, 0
        pl = oracle_lib.generate_output(q, ramp)[1][0, 0]
        assert (np.diff(pl) >= -1e-6).all(), tm


def test_oracle_output_resize_is_bilinear():
    w, h = 64, 32
    fm = _synthetic_moment(w, h, 1, seed=3)
    p = default_output_params(w, h, 1, dst=(2 * w - 1, 2 * h - 1))
    p.dithering, p.tonemap, p.purkinje = 0, 0, 0
    big = _unpack(oracle_lib.generate_output(p, fm)[0])
    p2 = default_output_params(w, h, 1)
    p2.dithering, p2.tonemap, p2.purkinje = 0, 0, 0
    small = _unpack(oracle_lib.generate_output(p2, fm)[0])
    # destination pixels that coincide with source pixels carry the source value
    assert np.abs(big[::2, ::2] - small).max() <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("tonemap,filt,dither,dst,extras", [
    (4, 0, 1, None, {}), (0, 0, 0, None, {"purkinje": 0}), (1, 1, 1, None, {}), (2, 2, 0, None, {}), (3, 3, 1, None, {}),
    (5, 4, 1, None, {"exposure": 1.7}), (6, 5, 1, None, {"agx_slope": 1.1, "agx_power": 1.2, "agx_saturation": 0.8}),
    (4, 6, 1, (200, 77), {}), (4, 0, 1, (97, 131), {"use_color_correction": 1, "cc_h": 0.2, "cc_s": -0.1, "cc_v": 0.05, "film_grain": 0.3}),
    (4, 0, 1, None, {"passthrough": 1}),
])
def test_output_chain_matches_oracle(tonemap, filt, dither, dst, extras):
    """Every tone curve, filter, the resize path, colour correction, film grain: ARGB8 bytes and float planes identical to the oracle."""
    from luminary_amd.core import Core
    core = Core(0)
    w, h, spp = 160, 90, 6
    fm = _synthetic_moment(w, h, spp, seed=5)
    p = default_output_params(w, h, spp, dst=dst)
    p.tonemap, p.filter, p.dithering = tonemap, filt, dither
    for k, v in extras.items():
        setattr(p, k, v)
    got, got_planes = core.generate_output(p, fm, want_float=True)
    want, want_planes = oracle_lib.generate_output(p, fm)
    assert np.array_equal(got_planes.view(np.uint32), want_planes.view(np.uint32)), "display-referred planes differ"
    assert np.array_equal(got, want), "%d of %d ARGB8 words differ" % ((got != want).sum(), got.size)
    assert got.max() > 0xFF000000


# ---- supersampling and the undersampling preview (generate_final_image / convert_RGBF_to_ARGB8 with their scales) ----
def test_oracle_supersampling_is_a_box_filter_of_tone_mapped_pixels():
    w, h, spp = 48, 20, 3
    fm = _synthetic_moment(w, h, spp, seed=2)
    mean = (fm / np.float32(spp)).reshape(3, h, w)
    for ss in (1, 2):
        p = default_output_params(w, h, spp, supersampling=ss)
        p.dithering = p.tonemap = p.purkinje = 0
        assert (p.dst_width, p.dst_height) == (w >> ss, h >> ss)
        argb, planes = oracle_lib.generate_output(p, fm)
        assert planes.shape == (3, h >> ss, w >> ss) and argb.shape == (h >> ss, w >> ss)
        k = 1 << ss
        box = mean.reshape(3, h >> ss, k, w >> ss, k).astype(np.float64).mean(axis=(2, 4))
        assert np.allclose(planes, box, rtol=2e-6)
        # the full chain: identical to tone-mapping every rendered pixel, then averaging (not the other way round)
        q = default_output_params(w, h, spp, supersampling=ss)
        q0 = default_output_params(w, h, spp)
        _, fine = oracle_lib.generate_output(q0, fm)
        _, coarse = oracle_lib.generate_output(q, fm)
        want = fine.reshape(3, h >> ss, k, w >> ss, k).astype(np.float64).mean(axis=(2, 4))
        assert np.allclose(coarse, want, rtol=2e-6, atol=1e-7)
    # a request of another size resamples the box-filtered image bilinearly: a constant frame stays constant
    p = default_output_params(w, h, 1, dst=(31, 17), supersampling=1)
    p.dithering = p.tonemap = p.purkinje = 0
    flat = np.full((3, w * h), 0.25, np.float32)
    img = _unpack(oracle_lib.generate_output(p, flat)[0])
    assert img.shape[:2] == (17, 31) and len(np.unique(img[..., 0])) == 1


def test_oracle_undersampling_preview_images():
    from luminary_amd.core import undersampling_schedule
    w, h = 37, 22   # not multiples of the block size
    assert undersampling_schedule(0) == [] and undersampling_schedule(1) == [(1, 3), (1, 2), (1, 1), (1, 0)]
    sched = undersampling_schedule(3)
    assert sched == [(3, 3), (3, 2), (3, 1), (3, 0), (2, 2), (2, 1), (2, 0), (1, 2), (1, 1), (1, 0)]
    seen = np.zeros(w * h, np.int32)
    frame = np.zeros((3, w * h), np.float32)
    value = np.random.default_rng(3).random((3, w * h), dtype=np.float32)
    for stage, it in sched:
        px = oracle_lib.undersampling_pixels(w, h, stage, it)
        seen[px] += 1
        frame[:, px] += value[:, px]
        img = oracle_lib.result_undersampled(frame, w, h, stage, it)
        assert img.shape == (3, h >> stage, w >> stage)
        # every block shows the mean of its 4 - it rendered pixels
        scale = 1 << stage
        for (bx, by) in [(0, 0), ((w >> stage) - 1, (h >> stage) - 1)]:
            pts = [(min(bx * scale + (0 if i & 1 else scale >> 1), w - 1), min(by * scale + (0 if i & 2 else scale >> 1), h - 1)) for i in range(it, 4)]
            want = np.float32(0.0)
            for (x, y) in pts:
                assert seen[x + y * w] == 1, "the preview only reads pixels that exist"
                want = want + value[0, x + y * w]
            assert img[0, by, bx] == np.float32(want * np.float32(1.0 / (4 - it)))
    assert (seen == 1).all(), "after the schedule every pixel holds exactly its first sample"
    # the display chain of a coarse image: every stored pixel covers 2^(stage - supersampling) output pixels
    stage = 2
    compact = np.arange(3 * (h >> stage) * (w >> stage), dtype=np.float32).reshape(3, -1) / 100.0
    p = default_output_params(w, h, 1, undersampling_stage=stage)
    p.dithering = p.tonemap = p.purkinje = 0
    argb, planes = oracle_lib.generate_output(p, compact)
    assert planes.shape == (3, h >> stage, w >> stage) and argb.shape == (h, w)
    assert (argb[:4, :4] == argb[0, 0]).all() and argb[0, 4] != argb[0, 0]
    # with supersampling 1 a stage-1 image is shown as it is, a stage-2 image doubled
    p1 = default_output_params(w * 2, h * 2, 1, supersampling=1, undersampling_stage=1)
    p1.dithering = p1.tonemap = p1.purkinje = 0
    src = np.random.default_rng(1).random((3, w * h), dtype=np.float32)
    argb1, planes1 = oracle_lib.generate_output(p1, src)
    assert planes1.shape == (3, h, w) and argb1.shape == (h, w) and np.array_equal(planes1.reshape(3, -1), src)


@pytest.mark.gpu
@pytest.mark.parametrize("ss,stage,dst", [(1, 0, None), (2, 0, None), (1, 0, (100, 37)), (3, 0, (16, 9)), (1, 2, None), (0, 2, None), (1, 3, (64, 64)), (0, 1, (50, 30)),
                                          (2, 1, None), (1, 1, None)])
def test_output_chain_scales_match_oracle(ss, stage, dst):
    """Supersampled frames and the coarse images of the undersampling preview through the display chain: bytes and planes identical."""
    from luminary_amd.core import Core
    core = Core(0)
    w, h, spp = 168, 88, 5
    fm = _synthetic_moment(w >> stage, h >> stage, spp, seed=7)
    p = default_output_params(w, h, spp, dst=dst, supersampling=ss, undersampling_stage=stage)
    p.film_grain = 0.2
    got, got_planes = core.generate_output(p, fm, want_float=True)
    want, want_planes = oracle_lib.generate_output(p, fm)
    assert got_planes.shape == want_planes.shape
    assert np.array_equal(got_planes.view(np.uint32), want_planes.view(np.uint32)), "display-referred planes differ"
    assert np.array_equal(got, want), "%d of %d ARGB8 words differ" % ((got != want).sum(), got.size)
    core.close()


# ---- bloom (device_post.c): mip chain of the result image blended back into it ----
def test_oracle_bloom_properties():
    w, h = 40, 24
    flat = np.full((3, h, w), 0.5, np.float32)
    out = oracle_lib.post_bloom(flat, w, h, 0.25)
    # the tent filter sums to 16/20 and every level adds to the one above, so a constant image does not keep its level exactly: with four
    # levels the glow is 0.25 / 4 * 0.8 * (1 + 0.8 + 0.64) of it, the base 0.75 of it; the border reads zeros and is dimmer
    assert 0.40 < out[0, h // 2, w // 2] < 0.45 and out[0, 0, 0] < out[0, h // 2, w // 2]
    spot = np.zeros((3, h, w), np.float32)
    spot[:, 12, 20] = 100.0
    glow = oracle_lib.post_bloom(spot, w, h, 0.1)
    assert glow[0, 12, 20] < 100.0 and glow[0, 12, 23] > 0.0 and glow[0, 3, 3] > 0.0, "light spreads over the frame"
    assert glow[0, 12, 23] > glow[0, 12, 30] > 0.0
    assert np.array_equal(oracle_lib.post_bloom(spot, w, h, 0.0)[0, 12, 20:21], spot[0, 12, 20:21]), "blend 0: base * 1 + nothing"
    # too coarse for a chain: floor(log2(24)) = 4 levels, stage 3 leaves one -> untouched
    small = np.random.default_rng(0).random((3, h >> 3, w >> 3), dtype=np.float32)
    assert np.array_equal(oracle_lib.post_bloom(small, w, h, 0.3, stage=3), small)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,stage,blend", [(160, 90, 0, 0.01), (97, 61, 0, 0.3), (64, 64, 0, 1.0), (168, 88, 1, 0.05), (168, 88, 2, 0.2), (33, 17, 0, 0.1), (16, 9, 0, 0.5)])
def test_bloom_matches_oracle(w, h, stage, blend):
    """Every level of the chain down to 1 x 1 (whose coordinates are NaN by construction), odd sizes, undersampled images."""
    from luminary_amd.core import Core
    core = Core(0)
    img = _synthetic_moment(w >> stage, h >> stage, 1, seed=9).reshape(3, h >> stage, w >> stage)
    got = core.post_bloom(img, w, h, blend, stage)
    want = oracle_lib.post_bloom(img, w, h, blend, stage)
    assert np.isfinite(want).all()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "%d of %d values differ, max %g" % ((got != want).sum(), got.size, np.abs(got - want).max())
    core.close()
