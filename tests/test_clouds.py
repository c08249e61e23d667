"""Clouds (cuda/cloud.cuh, cloud_utils.cuh, cloud_noise.cuh, cloud_shadow.cuh, device_cloud.c): three layers of noise-density clouds marched
in sky mode DEFAULT, their shadow in the sky's in-scattering march. CPU: the oracle against independent properties; GPU: HIP == oracle bit for bit,
including the generated noise textures."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from luminary_amd import SKY_MODE_DEFAULT, SKY_MODE_HDRI, scenes
from test_particles import _view

L = oracle_lib.lib()
L.oracle_probe_cloud_noise.argtypes = [C.c_uint32, C.c_void_p, C.c_float, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
EARTH = 6371.0


def _with_clouds(host, **kw):
    c = host.get_cloud()
    c.active = True
    for k, v in kw.items():
        if "." in k:
            layer, field = k.split(".")
            assert hasattr(getattr(c, layer), field), k
            setattr(getattr(c, layer), field, v)
        else:
            assert hasattr(c, k), k
            setattr(c, k, v)
    host.set_cloud(c)
    return host


def _cloud_view(host):
    return oracle_lib.with_cloud_noise(_view(host))


def _sky_scene(width=48, height=32, bounces=3, pitch=0.45, **sky):
    """The material zoo under the procedural sky with the camera tilted up: most of the frame is sky."""
    host = scenes.zoo_scene(width, height, bounces, sky_mode=SKY_MODE_DEFAULT)
    scenes.set_camera(host, (0.5, 3.2, 13.0), (pitch, 0.03, 0.0), fov=0.9)
    if sky:
        k = host.get_sky()
        for name, v in sky.items():
            assert hasattr(k, name), name
            setattr(k, name, v)
        host.set_sky(k)
    return host


# ---------------------------------------------------------------- CPU

def test_noise_textures_tile_and_fill_their_range():
    """The textures are sampled with wrap addressing: the Perlin and Worley octaves are built to tile (cloud_noise.cuh:21-60, :144-156), so the step across the
    wrap boundary is of the size of the step between any two neighbouring slices - up to a factor two: the Perlin hash reduces cell indices modulo 69
    BEFORE it wraps the neighbour index at the octave's scale (:27-37), so octaves of more than 69 cells (the four finest of seven) do not close exactly
    (kept). Every channel uses most of its 8 bits."""
    shape, detail, weather = oracle_lib.cloud_noise(1)
    for name, tex, dims in (("shape", shape, (128, 128, 128)), ("detail", detail, (32, 32, 32)), ("weather", weather, (1024, 1024))):
        t = tex.view(np.uint8).reshape(dims + (4,)).astype(np.float64)
        for axis in range(len(dims)):
            inner = np.abs(np.diff(t, axis=axis)).mean(axis=tuple(range(len(dims))))
            first, last = np.take(t, 0, axis=axis), np.take(t, dims[axis] - 1, axis=axis)
            seam = np.abs(first - last).mean(axis=tuple(range(len(dims) - 1)))
            ok = seam < 2.5 * inner + 0.5
            if name == "detail":
                ok[1] = True   # its second channel is Worley noise of scale 15 x 0.5 = 7.5 cells per texture: not an integer, so that one does not tile (kept)
            assert ok.all(), (name, axis, seam, inner)
        channels = 3 if name == "detail" else 4
        spread = t[..., :channels].reshape(-1, channels).max(axis=0) - t[..., :channels].reshape(-1, channels).min(axis=0)
        assert (spread > 100).all(), (name, spread)
    assert (detail.view(np.uint8).reshape(-1, 4)[:, 3] == 255).all()
    other = oracle_lib.cloud_noise(7)
    assert np.array_equal(other[0], shape) and np.array_equal(other[1], detail), "only the weather map depends on the seed (device_cloud.c:93-97)"
    assert (other[2] != weather).mean() > 0.3


def test_noise_octaves_close_at_the_wrap_and_are_bounded():
    """Worley octaves hash their cells modulo the scale: periodic with period 1 in texture coordinates away from the lower faces. The Perlin octaves wrap only the neighbour index of
    their last cell (cloud_noise.cuh:34-37): continuous across the texture's edge, not periodic beyond it."""
    rng = np.random.RandomState(3)
    p = rng.rand(2000, 3).astype(np.float32)

    def octaves(pts):
        pts = np.ascontiguousarray(pts, dtype=np.float32)
        per, wor = np.zeros(len(pts), np.float32), np.zeros(len(pts), np.float32)
        L.oracle_probe_cloud_noise(len(pts), pts.ctypes.data, 4.0, 3, 0.0, 0.3, per.ctypes.data, wor.ctypes.data)
        return per, wor

    per, wor = octaves(p)
    shifted = np.abs(wor - octaves(p + np.array([1.0, 0.0, 1.0], np.float32))[1])
    # (fmodf keeps the sign: the cells at index -1 next to the lower faces hash differently from the cells at scale - 1 they should repeat - kept)
    assert np.median(shifted) < 1e-5 and (shifted > 1e-3).mean() < 0.25, (np.median(shifted), (shifted > 1e-3).mean())
    lo, hi = p.copy(), p.copy()
    lo[:, 0], hi[:, 0] = 1e-4, 1.0 - 1e-4
    step = np.abs(octaves(lo)[0] - octaves(hi)[0])
    inside = np.abs(octaves(p)[0] - octaves(p + np.array([2e-4, 0.0, 0.0], np.float32))[0])
    assert step.max() < 0.03 and step.mean() < 3 * inside.mean() + 1e-3, (step.max(), step.mean(), inside.mean())
    assert 0.0 < per.min() and per.max() < 1.75 and 0.6 < per.mean() < 1.1, "three octaves of a [0, 1] Perlin value with weights 1, 1/2, 1/4"
    assert -0.6 < wor.min() and wor.max() <= 1.0, "inverted nearest-feature distance minus two finer octaves"


def test_density_lives_inside_the_layers_and_follows_coverage():
    host = _with_clouds(_sky_scene())
    view = _cloud_view(host)
    rng = np.random.RandomState(5)
    n = 4000
    pos = np.stack([rng.uniform(-60, 60, n), EARTH + rng.uniform(0.5, 9.0, n), rng.uniform(-60, 60, n)], axis=1).astype(np.float32)
    res = {}
    for layer in range(3):
        h, d = np.zeros(n, np.float32), np.zeros(n, np.float32)
        L.oracle_probe_cloud_density(C.byref(view), C.c_int(layer), C.c_uint32(n), pos.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p))
        res[layer] = (h, d)
        lo, hi = view.cloud_layers[layer][2], view.cloud_layers[layer][1]
        alt = np.linalg.norm(pos.astype(np.float64), axis=1) - EARTH
        assert np.allclose(h, (alt - lo) / (hi - lo), atol=2e-3 / (hi - lo))
        assert (d[(h < 0) | (h > 1)] == 0).all() and (d >= 0).all() and d.max() <= 1.0
    assert (res[0][1] > 0).mean() > 0.005, "the default low layer is partly covered"
    thin = _with_clouds(_sky_scene(), **{"low.coverage": 0.02})
    v2 = _cloud_view(thin)
    h, d = np.zeros(n, np.float32), np.zeros(n, np.float32)
    L.oracle_probe_cloud_density(C.byref(v2), C.c_int(0), C.c_uint32(n), pos.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p))
    assert (d == 0).all(), "coverage below the weather cut-off: no significant point (cloud_utils.cuh:405-423)"


def test_vanishing_clouds_leave_the_image():
    """Clouds of density 1e-7 with atmosphere_scattering off: the march runs (steps, random numbers, both light marches) but what it adds and removes is
    below float resolution of the sky behind it."""
    host = _sky_scene()
    base, _, cnt0 = oracle_lib.render(_view(host), 0, 2)
    _with_clouds(host, density=1e-7, atmosphere_scattering=False)
    fm, _, cnt1 = oracle_lib.render(_cloud_view(host), 0, 2)
    assert np.isfinite(fm).all()
    assert np.allclose(fm, base, rtol=2e-4, atol=1e-6)
    assert list(cnt0[:4]) == list(cnt1[:4])


def test_clouds_change_the_sky_and_shade_the_ground():
    host = _sky_scene(64, 40, 3, pitch=0.3)
    base = oracle_lib.render(_view(host), 0, 2)[0].reshape(3, 40, 64)
    _with_clouds(host)
    fm = oracle_lib.render(_cloud_view(host), 0, 2)[0].reshape(3, 40, 64)
    assert np.isfinite(fm).all() and (fm >= 0).all()
    sky_rows = slice(0, 12)
    assert (fm[:, sky_rows] != base[:, sky_rows]).mean() > 0.9
    # an overcast sky from below: darker than the clear sky towards the zenith but not black (the clouds scatter sun and sky light downwards)
    ratio = fm[:, sky_rows].mean() / base[:, sky_rows].mean()
    assert 0.05 < ratio < 3.0, ratio
    # without atmosphere_scattering the path's origin is not moved and the air in front of the clouds is not marched here
    _with_clouds(host, atmosphere_scattering=False)
    fm2 = oracle_lib.render(_cloud_view(host), 0, 2)[0].reshape(3, 40, 64)
    assert (fm2[:, sky_rows] != fm[:, sky_rows]).mean() > 0.5


# ---------------------------------------------------------------- GPU: HIP == oracle

def _parity(host, samples=2, spp_pass=2, counters=4, view=None):
    from luminary_amd.core import Core
    view = view or _cloud_view(host)
    core = Core(0)
    try:
        assert core.flavour == "exact"
        core.upload(view)
        core.set_pixels(None)
        core.reset_counters()
        core.render(0, samples, samples_per_pass=spp_pass)
        fm, sm = core.accumulators()
        ofm, osm, ocnt = oracle_lib.render(view, 0, samples)
        assert np.isfinite(ofm).all()
        assert np.array_equal(fm, ofm), "first moment: %d of %d differ, max %g" % ((fm != ofm).sum(), fm.size, np.abs(fm - ofm).max())
        assert np.array_equal(sm, osm)
        assert core.query_counters()[:counters] == [int(x) for x in ocnt[:counters]], (core.query_counters()[:4], list(ocnt[:4]))
        return ofm
    finally:
        core.close()


@pytest.mark.gpu
def test_generated_noise_textures_match_the_oracle():
    from luminary_amd.core import Core
    core = Core(0)
    try:
        for seed in (1, 42):
            got = core.cloud_noise_generate(seed)
            want = oracle_lib.cloud_noise(seed)
            for name, g, w in zip(("shape", "detail", "weather"), got, want):
                assert np.array_equal(g, w), "%s (seed %d): %d of %d texels differ" % (name, seed, (g != w).sum(), g.size)
    finally:
        core.close()


@pytest.mark.gpu
@pytest.mark.parametrize("options", [dict(), dict(atmosphere_scattering=False), dict(seed=9, offset_x=31.0, offset_z=-12.0, density=0.5),
                                     {"mid.active": False, "top.active": False, "low.type": 0.4, "low.coverage": 0.8, "low.wind_speed": 6.0, "low.wind_angle": 1.1},
                                     {"low.active": False, "steps": 64, "shadow_steps": 5, "octaves": 4, "droplet_diameter": 12.0},
                                     {"low.coverage_min": 0.6, "mid.type_min": 0.5, "noise_shape_scale": 1.7, "noise_detail_scale": 0.6, "noise_weather_scale": 2.5}])
def test_clouds_over_the_zoo_match_the_oracle(options):
    """The default three layers, without the air between them, another weather map and position, the low layer alone with wind, the upper layers alone with
    other step and octave counts and droplet size, other coverages and noise scales."""
    host = _with_clouds(_sky_scene(56, 36, 3), **options)
    fm = _parity(host)
    assert not np.array_equal(fm, oracle_lib.render(_view(_sky_scene(56, 36, 3)), 0, 2)[0])


@pytest.mark.gpu
def test_low_sun_and_a_camera_above_the_clouds_match_the_oracle():
    host = _with_clouds(_sky_scene(56, 36, 3, altitude=0.08, azimuth=2.2))
    _parity(host)
    host = _with_clouds(scenes.edge_scene("empty", 56, 36, 2))
    k = host.get_sky(); k.mode = SKY_MODE_DEFAULT; host.set_sky(k)
    scenes.set_camera(host, (0.0, 6500.0, 0.0), (-0.35, 0.4, 0.0), fov=1.0)   # 6.5 km up: between the mid and the top layer, looking down onto the low layer
    _parity(host)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [SKY_MODE_DEFAULT, SKY_MODE_HDRI])
def test_cloud_shadows_in_the_aerial_perspective_match_the_oracle(mode):
    """sky_trace_inscattering asks the clouds for a shadow (sky.cuh:374-376) whenever they are active - in HDRI mode too, where they are not marched. The
    scene is the kilometre-sized one of tests/test_sky.py: kilometres of air in front of every hit."""
    from test_sky import _distant_scene
    host = _distant_scene(True, mode=mode)
    plain = oracle_lib.render(_view(host), 0, 2)[0]
    _with_clouds(host, **{"low.coverage_min": 0.5, "low.height_min": 0.4, "low.height_max": 2.0})
    fm = _parity(host)
    assert not np.array_equal(fm, plain)


@pytest.mark.gpu
@pytest.mark.parametrize("shading_mode", [1, 2, 4])
def test_debug_modes_keep_the_aerial_perspective(shading_mode):
    """The debug queue keeps the in-scattering events (device_renderer.c:150-154): the air's light is added to every debug colour."""
    from test_sky import _distant_scene
    host = _distant_scene(True)
    st = host.get_settings(); st.shading_mode = shading_mode; host.set_settings(st)
    fm = _parity(host, counters=3, view=_view(host))
    clear = _distant_scene(False)
    clear.set_settings(st)
    assert not np.array_equal(fm, oracle_lib.render(_view(clear), 0, 2)[0])


def test_clouds_are_baked_into_the_panorama_cpu():
    """sky_compute_hdri with active clouds (sky_hdri.cuh:85-99): the panorama's colour changes where clouds are, its fourth channel is their transmittance -
    1 where the sky is clear or below the horizon's ground, less under clouds - and dims the sampled sun (sky_utils.cuh:343-346)."""
    host = scenes.zoo_scene(40, 28, 2, sky_mode=SKY_MODE_HDRI)
    clear = _view(host)
    pano0 = clear._hdri_keep
    assert (pano0[..., 3] == 1.0).all()
    _with_clouds(host, **{"low.coverage_min": 0.3})
    cloudy = _cloud_view(host)
    pano1 = cloudy._hdri_keep
    assert pano1.shape == pano0.shape and np.isfinite(pano1).all()
    upper = slice(0, pano1.shape[0] // 2 - 2)
    assert (pano1[upper, :, 3] < 0.999).mean() > 0.2 and pano1[..., 3].min() >= 0.0 and pano1[..., 3].max() <= 1.0
    assert (pano1[upper, :, :3] != pano0[upper, :, :3]).mean() > 0.5
    a = oracle_lib.render(clear, 0, 2)[0]
    b = oracle_lib.render(cloudy, 0, 2)[0]
    assert np.isfinite(b).all() and not np.array_equal(a, b)


@pytest.mark.gpu
def test_clouds_baked_into_the_panorama_match_the_oracle():
    from luminary_amd.core import Core
    host = scenes.zoo_scene(48, 32, 3, sky_mode=SKY_MODE_HDRI)
    k = host.get_sky(); k.hdri_dim, k.hdri_samples = 24, 3; host.set_sky(k)
    _with_clouds(host, **{"low.coverage_min": 0.3})
    oracle_view = oracle_lib.with_cloud_noise(oracle_lib.with_sky_hdri(host.device_scene()))   # the oracle's bake, clouds included
    gpu_view = oracle_lib.with_cloud_noise(oracle_lib.with_luts(host.device_scene()))           # no tables, no panorama: made on the GPU at upload
    core = Core(0)
    try:
        core.upload(gpu_view)
        got = core.sky_hdri_download()
        assert (got[..., 3] < 0.999).any()
        assert np.array_equal(got, oracle_view._hdri_keep), "the bake at upload: %d of %d values differ" % ((got != oracle_view._hdri_keep).sum(), got.size)
        core.set_pixels(None)
        core.reset_counters()
        core.render(0, 3, samples_per_pass=2)
        fm, sm = core.accumulators()
        ofm, osm, ocnt = oracle_lib.render(oracle_view, 0, 3)
        assert np.array_equal(fm, ofm), "first moment: %d of %d differ, max %g" % ((fm != ofm).sum(), fm.size, np.abs(fm - ofm).max())
        assert np.array_equal(sm, osm) and core.query_counters()[:4] == [int(x) for x in ocnt[:4]]
    finally:
        core.close()


@pytest.mark.gpu
def test_clouds_through_the_host_api():
    """luminary_host_set_cloud, then the library's own render entry: the core generates the noise textures itself (no pointers in the view)."""
    host = _with_clouds(_sky_scene(48, 32, 2), seed=3)
    assert host.get_cloud().active and host.get_cloud().seed == 3
    host.render_samples(0, 2)
    fm, sm = host.accumulators()
    ofm, osm, _ = oracle_lib.render(_cloud_view(host), 0, 2)
    assert np.array_equal(fm, ofm) and np.array_equal(sm, osm)


@pytest.mark.gpu
def test_fast_flavour_renders_the_same_clouds():
    """The default (fast) flavour marches the same clouds (the noise textures exist once, made by the exact kernels) at 64 spp: same estimator, different
    rounding; a 3-way tile partition reproduces the frame bit for bit."""
    from luminary_amd.core import Core
    from luminary_amd.distributed import tile_pixels
    host = _with_clouds(_sky_scene(48, 32, 3))
    view = _cloud_view(host)
    frames = {}
    for flavour in ("exact", "fast"):
        core = Core(0)
        try:
            core.set_flavour(flavour)
            core.upload(view)
            core.set_pixels(None)
            core.render(0, 64, samples_per_pass=8)
            frames[flavour] = core.accumulators()[0].astype(np.float64) / 64
            if flavour == "fast":
                core.set_pixels(None)
                core.render(0, 2, samples_per_pass=2)
                full = core.accumulators()[0]
                acc = np.zeros_like(full)
                for rank in range(3):
                    tiles = tile_pixels(48, 32, rank, 3, tile=16)
                    core.set_pixels(tiles)
                    core.render(0, 2, samples_per_pass=2)
                    acc[:, tiles] = core.accumulators()[0]
                assert np.array_equal(acc, full)
        finally:
            core.close()
    a, b = frames["exact"], frames["fast"]
    assert np.isfinite(b).all()
    rel_l2 = np.linalg.norm(a - b) / np.linalg.norm(a)
    assert rel_l2 < 0.05, rel_l2
    assert abs(b.sum() / a.sum() - 1.0) < 5e-3, b.sum() / a.sum()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [SKY_MODE_DEFAULT, SKY_MODE_HDRI])
def test_everything_at_once_matches_the_oracle(mode):
    """Ocean, fog, particles, clouds and aerial perspective in one frame - every kernel of the schedule in one pass, in the procedural mode (clouds marched
    per path) and in the panorama mode (clouds baked in, their transmittance dimming the sun)."""
    from test_particles import _with_particles
    host = scenes.zoo_scene(48, 32, 4, sky_mode=mode)
    scenes.set_camera(host, (0.5, 3.2, 13.0), (0.12, 0.03, 0.0), fov=0.9)
    k = host.get_sky(); k.aerial_perspective = True; k.hdri_dim, k.hdri_samples = 16, 2; host.set_sky(k)
    o = host.get_ocean(); o.active, o.height, o.amplitude, o.frequency = True, 1.2, 0.3, 0.5; host.set_ocean(o)
    f = host.get_fog(); f.active, f.density = True, 50.0; host.set_fog(f)
    _with_particles(host, count=1500, size=20.0, scale=5.0)
    _with_clouds(host)
    _parity(host, samples=3)
