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
