"""The ocean (cuda/ocean_utils.cuh, cuda/ocean.cuh, cuda/caustics.cuh, direct_lighting.cuh:123-243, :466-584): a procedural height field
that is ray-marched, the water below it as the second volume type, the sun and the ambient sample seen through the surface. CPU: the oracle against
independent numpy; GPU: HIP == oracle bit for bit."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from luminary_amd import SKY_MODE_CONSTANT_COLOR, SKY_MODE_DEFAULT, SKY_MODE_HDRI, scenes
from test_particles import _squares16, _view, _with_particles

L = oracle_lib.lib()
L.oracle_probe_ocean_fresnel.argtypes = [C.c_float, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]


def _with_ocean(host, height=1.0, amplitude=0.3, frequency=0.5, **kw):
    o = host.get_ocean()
    o.active, o.height, o.amplitude, o.frequency = True, height, amplitude, frequency
    for k, v in kw.items():
        assert hasattr(o, k), k
        setattr(o, k, v)
    host.set_ocean(o)
    return host


# ---------------------------------------------------------------- CPU: the oracle against independent mathematics

def _np_height(view, x, z):
    """ocean_get_height (ocean_utils.cuh:26-111) in numpy: float32 where the arithmetic decides an integer (the lattice hash), numpy's own sin / cos."""
    f32 = np.float32

    def hash_(ix, iy):
        v = np.abs(ix + iy * f32(311.7 / 127.1)).astype(f32)
        off = np.where(v < 4294967040.0, v, 4294967295.0).astype(np.uint64).astype(np.uint32)
        return (_squares16(off).astype(np.float64) * 2 ** -16).astype(f32)   # 0x3F800000 | v << 7, minus 1

    def noise(px, py):
        ix, iy = np.floor(px), np.floor(py)
        fx, fy = px - ix, py - iy
        fx = fx * fx * (f32(3) - f32(2) * fx); fy = fy * fy * (f32(3) - f32(2) * fy)
        h1, h2, h4, h3 = hash_(ix, iy), hash_(ix + 1, iy), hash_(ix + 1, iy + 1), hash_(ix, iy + 1)
        a = h1 + (h2 - h1) * fx; b = h3 + (h4 - h3) * fx
        return -1 + 2 * (a + (b - a) * fy)

    def octave(px, py):
        off = noise(px, py)
        px, py = px + off, py + off
        w1x, w1y = 1 - np.abs(np.sin(px)), 1 - np.abs(np.sin(py))
        w2x, w2y = np.abs(np.cos(px)), np.abs(np.cos(py))
        w1x = w1x + (w2x - w1x) * w1x; w1y = w1y + (w2y - w1y) * w1y
        return (1 - np.sqrt(w1x * w1y)) ** 2

    qx, qy = (x * f32(0.75)).astype(f32), z.astype(f32)
    amp, freq, h = f32(1), f32(view.ocean_frequency), np.zeros_like(qx)
    for _ in range(8):
        h = h + octave(qx * freq, qy * freq) * amp
        qx, qy = f32(1.6) * qx - f32(1.2) * qy, f32(1.2) * qx + f32(1.6) * qy
        freq, amp = freq * f32(1.9), amp * f32(0.22)
    return h * f32(view.ocean_amplitude)


def test_height_field_follows_the_reference_function(tmp_path):
    host = _with_ocean(scenes.cornell_host(str(tmp_path), 8, 8, 1), height=0.0, amplitude=0.6, frequency=0.16)
    view = host.device_scene()
    assert view.ocean_active == 1 and view.ocean_amplitude == np.float32(0.6)
    rng = np.random.RandomState(4)
    xz = rng.uniform(-40, 40, size=(4000, 2)).astype(np.float32)
    out = np.zeros(4000, np.float32)
    L.oracle_probe_ocean_height(C.byref(view), C.c_uint32(4000), xz.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    want = _np_height(view, xz[:, 0], xz[:, 1])
    assert np.abs(out - want).max() < 2e-4, np.abs(out - want).max()
    assert 0.0 <= out.min() and out.max() <= 1.33 * 0.6, "OCEAN_MAX_HEIGHT bounds the field (ocean_utils.cuh:19-21)"
    assert out.std() > 0.02


def test_ray_marcher_ends_on_the_surface_and_on_its_first_crossing(tmp_path):
    """ocean_intersection_distance (ocean_utils.cuh:161-287) against a dense march of the numpy height field: the end point lies on the surface (residual
    below the solver's 1e-4 target, or inside its last bracket), and for nearly all rays no crossing lies before it (the marcher's step rule is an
    approximate Lipschitz bound: a thin crest may be stepped over - reference behaviour, so not every ray is required)."""
    host = _with_ocean(scenes.cornell_host(str(tmp_path), 8, 8, 1), height=1.0, amplitude=0.5, frequency=0.3)
    view = host.device_scene()
    rng = np.random.RandomState(9)
    n = 300
    o = np.stack([rng.uniform(-20, 20, n), rng.uniform(2.5, 6.0, n), rng.uniform(-20, 20, n)], axis=1).astype(np.float32)
    d = np.stack([rng.normal(size=n), -rng.uniform(0.15, 1.0, n), rng.normal(size=n)], axis=1)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    o[n // 2:, 1] = rng.uniform(-3.0, 0.5, n - n // 2); d[n // 2:, 1] *= -1   # the second half starts under water and looks up
    lim = np.full(n, 3.0e38, np.float32)
    t = np.zeros(n, np.float32); res = np.zeros(n, np.float32); nrm = np.zeros((n, 3), np.float32)
    L.oracle_probe_ocean_trace(C.byref(view), C.c_uint32(n), o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), lim.ctypes.data_as(C.c_void_p),
                               t.ctypes.data_as(C.c_void_p), res.ctypes.data_as(C.c_void_p), nrm.ctypes.data_as(C.c_void_p))
    hit = t < 1e30
    assert hit.mean() > 0.95
    assert np.percentile(np.abs(res[hit]), 90) < 1.5e-4 and np.abs(res[hit]).max() < 0.2, "the rest: midpoint of the last bracket (ocean_utils.cuh:262-264)"
    first_ok = 0
    for r in np.nonzero(hit)[0]:
        s = np.linspace(0.0, float(t[r]) * 0.995, 600)
        p = o[r].astype(np.float64)[None] + s[:, None] * d[r].astype(np.float64)[None]
        rel = p[:, 1] - (1.0 + _np_height(view, p[:, 0].astype(np.float32), p[:, 2].astype(np.float32)))
        inside = (p[:, 1] >= 1.0) & (p[:, 1] <= 1.0 + 1.33 * 0.5)
        first_ok += not (np.sign(rel[inside][:-1]) * np.sign(rel[inside][1:]) < 0).any() if inside.sum() > 1 else 1
    assert first_ok >= 0.97 * hit.sum(), (first_ok, hit.sum())
    # the normal is the gradient of the field: central differences of the numpy height at a step far above the Sobel filter's (whose step of a few
    # float32 ulps of the position makes the reference's normal noisy by a few degrees)
    p = o[hit] + t[hit, None] * d[hit]
    e = np.float32(2e-3)
    gx = (_np_height(view, p[:, 0] + e, p[:, 2]) - _np_height(view, p[:, 0] - e, p[:, 2])) / (2 * e)
    gz = (_np_height(view, p[:, 0], p[:, 2] + e) - _np_height(view, p[:, 0], p[:, 2] - e)) / (2 * e)
    g = np.stack([-gx, np.ones_like(gx), -gz], axis=1); g /= np.linalg.norm(g, axis=1, keepdims=True)
    cosine = (g * nrm[hit]).sum(axis=1)
    assert np.median(cosine) > 0.995 and np.percentile(cosine, 10) > 0.97, (np.median(cosine), np.percentile(cosine, 10))


def test_flat_water_and_rays_that_miss(tmp_path):
    host = _with_ocean(scenes.cornell_host(str(tmp_path), 8, 8, 1), height=2.0, amplitude=0.0)
    view = host.device_scene()
    o = np.array([[0, 5, 0], [0, 5, 0], [0, -1, 0], [3, 2.0, 1], [0, 5, 0]], np.float32)
    d = np.array([[0, -1, 0], [0, 1, 0], [0.6, 0.8, 0], [1, 0, 0], [0.8, -0.6, 0]], np.float32)
    lim = np.array([3e38, 3e38, 3e38, 3e38, 4.0], np.float32)
    t = np.zeros(5, np.float32); res = np.zeros(5, np.float32); nrm = np.zeros((5, 3), np.float32)
    L.oracle_probe_ocean_trace(C.byref(view), C.c_uint32(5), o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), lim.ctypes.data_as(C.c_void_p),
                               t.ctypes.data_as(C.c_void_p), res.ctypes.data_as(C.c_void_p), nrm.ctypes.data_as(C.c_void_p))
    assert t[0] == 3.0 and t[1] > 1e30 and abs(t[2] - 3.75) < 1e-6, t
    assert t[3] == 0.0, "a ray that starts in the surface's slab starts the march at 0"
    assert t[4] == 5.0, "a flat surface does not look at the limit (ocean_utils.cuh:279-281); the caller compares"
    assert np.array_equal(nrm[0], [0, 1, 0])


def test_reflection_coefficient_is_fresnel(tmp_path):
    """ocean_reflection_coefficient (ocean_utils.cuh:457-475) against the Fresnel equations for unpolarised light, refraction against Snell's law."""
    th = np.linspace(0.0, 1.55, 64)
    d = np.stack([np.sin(th), -np.cos(th), np.zeros_like(th)], axis=1).astype(np.float32)
    R = np.zeros(64, np.float32); T = np.zeros((64, 3), np.float32)
    L.oracle_probe_ocean_fresnel(1.333, 64, d.ctypes.data, R.ctypes.data, T.ctypes.data)
    tt = np.arcsin(np.sin(th) / 1.333)
    rs = ((np.cos(th) - 1.333 * np.cos(tt)) / (np.cos(th) + 1.333 * np.cos(tt))) ** 2
    rp = ((np.cos(tt) - 1.333 * np.cos(th)) / (np.cos(tt) + 1.333 * np.cos(th))) ** 2
    assert np.abs(R - 0.5 * (rs + rp)).max() < 1e-5
    assert np.abs(T[:, 0] - np.sin(tt)).max() < 1e-5 and np.abs(T[:, 1] + np.cos(tt)).max() < 1e-5


def test_jerlov_water_types_reach_the_scene(tmp_path):
    """device_struct_ocean_convert (device_structs.c:87-105) with the coefficient tables of ocean_utils.cuh:291-385: clearer water types scatter and absorb
    less, red is absorbed most; the RIS sample count is stored minus one."""
    host = scenes.cornell_host(str(tmp_path), 8, 8, 1)
    rows = []
    for water_type in range(10):
        _with_ocean(host, water_type=water_type, caustics_ris_sample_count=32)
        v = host.device_scene()
        rows.append((list(v.ocean_scattering), list(v.ocean_absorption), v.ocean_molecular_weight))
        assert v.ocean_caustics_ris_sample_count == 31
    sc = np.array([r[0] for r in rows]); ab = np.array([r[1] for r in rows]); w = np.array([r[2] for r in rows])
    assert (sc > 0).all() and (ab > 0).all() and (0 <= w).all() and (w <= 1).all()
    assert (np.diff(sc[:5].sum(axis=1)) > 0).all() and (np.diff(ab[:5].sum(axis=1)) > 0).all(), "types I .. III: open ocean, increasingly turbid"
    assert (ab[:5, 0] > ab[:5, 2]).all(), "open ocean water absorbs red more than blue"
    assert (np.diff(w[:5]) < 0).all(), "the share of molecular scattering falls with turbidity"


def test_clear_flat_water_conserves_energy():
    """A furnace test the restatement has to pass by physics, not by construction: an empty scene under a constant white sky, flat water whose volume neither
    scatters nor absorbs. Whatever a camera ray does at the surface - reflect by Fresnel or refract and leave downwards - it ends in the sky, so
    every pixel that sees the water is white as well (the bounce weights sum to one)."""
    host = scenes.edge_scene("empty", 24, 16, 4)
    scenes.set_camera(host, (0.0, 3.0, 0.0), (-0.5, 0.0, 0.0))
    _with_ocean(host, height=0.0, amplitude=0.0)
    view = _view(host)
    for k in range(3):
        view.ocean_scattering[k] = 0.0; view.ocean_absorption[k] = 0.0
    sky = np.array(list(view.sky_constant_color))
    fm, _, cnt = oracle_lib.render(view, 0, 64)
    img = fm.reshape(3, 16, 24) / 64.0   # the first moment is a sum over the samples
    water = img[:, 12:, :]   # the lower rows look at the water
    assert np.abs(water.mean(axis=(1, 2)) / sky - 1.0).max() < 0.03, water.mean(axis=(1, 2)) / sky
    assert cnt[3] == 0, "no surface vertex in an empty scene"


def test_the_ocean_changes_what_is_seen_above_and_below(tmp_path):
    host = scenes.zoo_scene(40, 28, 4)
    base, _, cnt0 = oracle_lib.render(_view(host), 0, 3)
    _with_ocean(host, height=1.0)
    above, _, cnt1 = oracle_lib.render(_view(host), 0, 3)
    _with_ocean(host, height=4.5)
    below, _, cnt2 = oracle_lib.render(_view(host), 0, 3)
    assert np.isfinite(above).all() and np.isfinite(below).all()
    assert (above != base).mean() > 0.3 and (below != base).mean() > 0.9
    assert below.mean() < base.mean(), "the water absorbs"
    a, b = below.reshape(3, -1).mean(axis=1), base.reshape(3, -1).mean(axis=1)
    assert a[0] / b[0] < a[2] / b[2], "red is absorbed more than blue"


# ---------------------------------------------------------------- GPU: HIP == oracle

def _parity(host, samples=3, spp_pass=2, counters=4, edit=None):
    from luminary_amd.core import Core
    view = _view(host)
    if edit:
        edit(view)
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
@pytest.mark.parametrize("mode", [SKY_MODE_CONSTANT_COLOR, SKY_MODE_DEFAULT, SKY_MODE_HDRI])
@pytest.mark.parametrize("height", [1.0, 4.5])
def test_ocean_in_the_zoo_matches_the_oracle(mode, height):
    """The camera above the water (height 1: the lower half of the objects is submerged, paths enter and leave the water) and below it (height 4.5: every
    vertex takes the sun through the surface - caustics with the resampled patch - and its ambient sample in two segments)."""
    host = _with_ocean(scenes.zoo_scene(64, 40, 5, sky_mode=mode), height=height)
    fm = _parity(host, samples=3)
    plain = scenes.zoo_scene(64, 40, 5, sky_mode=mode)
    assert not np.array_equal(fm, oracle_lib.render(_view(plain), 0, 3)[0])


@pytest.mark.gpu
@pytest.mark.parametrize("options", [dict(caustics_active=False), dict(caustics_active=True, caustics_ris_sample_count=1), dict(amplitude=0.0),
                                     dict(multiscattering=True), dict(triangle_light_contribution=True), dict(water_type=7, refractive_index=1.5)])
def test_ocean_options_match_the_oracle(options):
    """Under water with the procedural sky: the caustics fast path, a single RIS sample, flat water, multiple scattering, bridges to the emissive
    triangles through the water, a turbid coastal water type with another refractive index."""
    kw = dict(height=4.5)
    kw.update(options)
    host = _with_ocean(scenes.zoo_scene(56, 36, 5, sky_mode=SKY_MODE_DEFAULT), **kw)
    _parity(host, samples=3)


@pytest.mark.gpu
def test_ocean_with_fog_and_particles_matches_the_oracle():
    """All three at once, camera above the water: the fog is the volume above the surface and the second volume of a vertex under water; particles are hit
    above and below."""
    host = _with_ocean(scenes.zoo_scene(56, 36, 5, sky_mode=SKY_MODE_HDRI), height=1.2, amplitude=0.4)
    f = host.get_fog(); f.active, f.density = True, 60.0; host.set_fog(f)
    _with_particles(host, count=2500, size=20.0, scale=5.0)
    _parity(host, samples=3)


@pytest.mark.gpu
def test_ocean_with_fog_under_water_matches_the_oracle():
    host = _with_ocean(scenes.zoo_scene(56, 36, 5, sky_mode=SKY_MODE_DEFAULT), height=4.5, triangle_light_contribution=True)
    f = host.get_fog(); f.active, f.density = True, 60.0; host.set_fog(f)
    _with_particles(host, count=2500, size=20.0, scale=5.0)
    _parity(host, samples=3)


@pytest.mark.gpu
@pytest.mark.parametrize("shading_mode", [1, 2, 3, 4])
def test_ocean_debug_modes_match_the_oracle(shading_mode):
    host = _with_ocean(scenes.zoo_scene(64, 40, 3), height=1.0)
    st = host.get_settings(); st.shading_mode = shading_mode; host.set_settings(st)
    _parity(host, samples=2, counters=3)


@pytest.mark.gpu
def test_empty_scene_with_an_ocean_and_clear_water():
    """No geometry: the surface, the sky and the water volume only; with zeroed coefficients the furnace test of the CPU section runs on the GPU."""
    host = scenes.edge_scene("empty", 48, 32, 4)
    scenes.set_camera(host, (0.0, 3.0, 0.0), (-0.5, 0.0, 0.0))
    _with_ocean(host, height=0.0, amplitude=0.5)
    _parity(host, samples=4, spp_pass=4)

    def clear(view):
        for k in range(3):
            view.ocean_scattering[k] = 0.0; view.ocean_absorption[k] = 0.0
    _parity(host, samples=4, spp_pass=4, edit=clear)


@pytest.mark.gpu
def test_ocean_through_the_host_api(tmp_path):
    """luminary_host_set_ocean, then the library's own render entry (luminary_ext_render_samples): its accumulators equal the oracle's."""
    host = scenes.cornell_host(str(tmp_path), 48, 32, 3)
    _with_ocean(host, height=0.4, amplitude=0.1, frequency=2.0)
    assert host.get_ocean().active and host.get_ocean().height == np.float32(0.4)
    host.render_samples(0, 2)
    fm, sm = host.accumulators()
    ofm, osm, _ = oracle_lib.render(_view(host), 0, 2)
    assert np.array_equal(fm, ofm) and np.array_equal(sm, osm)
    plain = scenes.cornell_host(str(tmp_path / "plain"), 48, 32, 3)
    assert not np.array_equal(ofm, oracle_lib.render(_view(plain), 0, 2)[0])


@pytest.mark.gpu
def test_fast_flavour_renders_the_same_ocean():
    """The default (fast) flavour under water (caustics, two-segment visibility, the water volume) at 128 spp against the exact one. The estimator is noisy
    here (caustics), so the yardstick is an exact render from other sample ids: the fast image is not further from the exact one than that is, and the frame
    sums agree as well as the two exact ones do."""
    from luminary_amd.core import Core
    host = _with_ocean(scenes.zoo_scene(48, 32, 4, sky_mode=SKY_MODE_DEFAULT), height=4.5)
    view = _view(host)
    frames = {}
    for name, flavour, first in (("exact", "exact", 0), ("other", "exact", 128), ("fast", "fast", 0)):
        core = Core(0)
        try:
            core.set_flavour(flavour)
            core.upload(view)
            core.set_pixels(None)
            core.render(first, 128, samples_per_pass=8)
            frames[name] = core.accumulators()[0].astype(np.float64) / 128
        finally:
            core.close()
    a, b, other = frames["exact"], frames["fast"], frames["other"]
    assert np.isfinite(b).all()
    noise = np.linalg.norm(a - other) / np.linalg.norm(a)
    rel_l2 = np.linalg.norm(a - b) / np.linalg.norm(a)
    assert rel_l2 < 1.1 * noise, (rel_l2, noise)
    assert abs(b.sum() / a.sum() - 1.0) < max(1e-2, 2.0 * abs(other.sum() / a.sum() - 1.0)), (b.sum() / a.sum(), other.sum() / a.sum())


@pytest.mark.gpu
def test_tile_partition_and_pass_shapes_with_an_ocean():
    """What the multi-GPU path and the batching rely on with the ocean's kernels in the loop: a 3-way tile partition of the frame reproduces the full frame bit for
    bit (every path owns its water factors and its 6 / 19 visibility slots), and so do two sample ids in one pass vs two passes."""
    from luminary_amd.core import Core
    from luminary_amd.distributed import tile_pixels
    host = _with_ocean(scenes.zoo_scene(96, 64, 4, sky_mode=SKY_MODE_HDRI), height=1.2)
    f = host.get_fog(); f.active, f.density = True, 50.0; host.set_fog(f)
    view = _view(host)
    core = Core(0)
    try:
        core.upload(view)
        core.set_pixels(None)
        core.render(0, 2, samples_per_pass=2)
        full, full_sm = core.accumulators()
        core.set_pixels(None)
        core.render(0, 1, samples_per_pass=1)
        core.render(1, 1, samples_per_pass=1)
        two, two_sm = core.accumulators()
        assert np.array_equal(two, full) and np.array_equal(two_sm, full_sm)
        acc, acc_sm = np.zeros_like(full), np.zeros_like(full_sm)
        for rank in range(3):
            tiles = tile_pixels(96, 64, rank, 3, tile=16)
            core.set_pixels(tiles)
            core.render(0, 2, samples_per_pass=2)
            part, part_sm = core.accumulators()
            acc[:, tiles] = part
            acc_sm[tiles] = part_sm
        assert np.array_equal(acc, full) and np.array_equal(acc_sm, full_sm)
    finally:
        core.close()


@pytest.mark.gpu
@pytest.mark.parametrize("flavour", ["exact", "fast"])
def test_a_thousand_samples_of_everything_at_once(flavour):
    """Soak: ocean + fog + particles + clouds under the procedural sky, 1024 sample ids through every kernel of the schedule. Rare samples produce non-finite
    directions (a grazing refraction, a degenerate in-scattering vertex); the traversals end such rays as misses (dev_trace.h) instead of walking off the
    tree, so the run completes and the image stays finite."""
    from luminary_amd.core import Core
    host = _with_ocean(scenes.zoo_scene(48, 32, 6, sky_mode=SKY_MODE_DEFAULT), height=1.5, amplitude=0.5, triangle_light_contribution=True, multiscattering=True)
    f = host.get_fog(); f.active, f.density = True, 40.0; host.set_fog(f)
    _with_particles(host, count=1500, size=20.0, scale=5.0)
    c = host.get_cloud(); c.active = True; host.set_cloud(c)
    k = host.get_sky(); k.aerial_perspective = True; host.set_sky(k)
    view = oracle_lib.with_cloud_noise(_view(host))
    core = Core(0)
    try:
        core.set_flavour(flavour)
        core.upload(view)
        core.set_pixels(None)
        core.render(0, 1024, samples_per_pass=16)
        fm, sm = core.accumulators()
    finally:
        core.close()
    assert np.isfinite(fm).mean() > 0.999 and (fm[np.isfinite(fm)] >= 0).all()
    assert fm[np.isfinite(fm)].mean() > 0.0


@pytest.mark.gpu
def test_adaptive_sampling_with_ocean_fog_and_clouds_matches_the_oracle():
    """The adaptive generator (k_generate_adaptive) starts its paths like k_generate: medium and volume stack of the camera position, then the whole schedule
    with the ocean's, the fog's and the clouds' kernels. Moments, per-block rates and the result image against the oracle's adaptive run."""
    from luminary_amd.core import Core, default_output_params
    W, H = 30, 22
    host = _with_ocean(scenes.zoo_scene(W, H, 3, sky_mode=SKY_MODE_DEFAULT), height=1.2)
    f = host.get_fog(); f.active, f.density = True, 40.0; host.set_fog(f)
    c = host.get_cloud(); c.active = True; host.set_cloud(c)
    view = oracle_lib.with_cloud_noise(_view(host))
    tone = default_output_params(W, H, 1)
    executions = 2 + 4 + 3
    o = oracle_lib.AdaptiveOracle(view, 6, 2, 2, exposure=0.0, tone=tone)
    o.render(executions)
    core = Core(0)
    try:
        core.upload(view)
        core.set_pixels(None)
        core.adaptive_begin(6, 2, 2, exposure=0.0, tone=tone)
        core.adaptive_render(executions)
        counts, variance = core.adaptive_download()
        assert np.array_equal(counts, o.stage_counts) and np.array_equal(variance, o.block_variance)
        fm, sm = core.accumulators()
        assert np.array_equal(fm, o.fm.reshape(3, -1)) and np.array_equal(sm, o.sm)
        assert np.array_equal(core.generate_result(mode=0, tone=tone), o.result(mode=0, tone=tone))
        core.adaptive_end()
    finally:
        core.close()
