"""The fog volume (cuda/volume.cuh + light_bridges.cuh): CPU checks of the oracle against independent mathematics, and GPU parity of
the HIP kernels (k_volume_inscatter / k_volume_resolve / k_volume_events / k_volume_bounce) with the oracle, bit for bit in the exact flavour."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib
from luminary_amd import SKY_MODE_CONSTANT_COLOR, SKY_MODE_DEFAULT, SKY_MODE_HDRI, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = oracle_lib.lib()


def _fogged(host, density=40.0, height=500.0, dist=500.0, diameter=10.0):
    f = host.get_fog()
    f.active, f.density, f.height, f.dist, f.droplet_diameter = True, density, height, dist, diameter
    host.set_fog(f)
    return host


def _view(host):
    sky = host.get_sky()
    if sky.mode == SKY_MODE_HDRI and sky.hdri_dim > 32:  # a small panorama: the oracle bakes it on the CPU
        sky.hdri_dim, sky.hdri_samples = 32, 3
        host.set_sky(sky)
    plain = host.device_scene()
    if sky.mode == SKY_MODE_HDRI:
        return oracle_lib.with_sky_hdri(plain)
    if sky.mode == SKY_MODE_DEFAULT:
        return oracle_lib.with_sky_luts(oracle_lib.with_luts(plain))
    return oracle_lib.with_luts(plain)


# ---------------------------------------------------------------- CPU: the oracle against independent mathematics

def test_random_targets_of_the_oracle_follow_the_allocation_rule():
    """Every RT_* / RANDOM_TARGET_* constant of the oracle against the table random.cuh:24-66 generates (START_next = START + count * sets + 1); the
    sets of the volume context are LIGHT_SUN<1>, LIGHT_GEO<1>, BSDF<0>/<2> (material.cuh:76-81). This check found SKY_INSCATTERING_STEP one too low."""
    rows = [("LENS_METHOD", 32, 1), ("LENS", 1, 1), ("LENS_BLADE", 1, 1), ("LENS_WAVELENGTH", 1, 1), ("BSDF_REFLECTION", 1, 3), ("BSDF_DIFFUSE", 1, 3),
            ("BSDF_REFRACTION", 1, 3), ("BSDF_RESAMPLING", 1, 3), ("BSDF_OPACITY", 1, 3), ("VOLUME_INTERSECTION", 1, 1), ("RUSSIAN_ROULETTE", 1, 1),
            ("CAMERA_JITTER", 1, 1), ("CAMERA_TIME", 1, 1), ("CLOUD_STEP_OFFSET", 3, 1), ("CLOUD_STEP_COUNT", 3, 1), ("CLOUD_DIR", 1, 1),
            ("SKY_STEP_OFFSET", 1, 1), ("SKY_INSCATTERING_STEP", 1, 1), ("CAUSTIC_INITIAL", 128, 2), ("CAUSTIC_RESAMPLING", 1, 2),
            ("CAUSTIC_SUN_RAY", 1, 2), ("LIGHT_SUN_INITIAL_VERTEX", 1, 1), ("LIGHT_SUN_BSDF", 1, 2), ("LIGHT_SUN_BSDF_METHOD", 1, 2),
            ("LIGHT_SUN_RAY", 1, 2), ("LIGHT_SUN_RESAMPLING", 1, 2), ("LIGHT_GEO_INITIAL_VERTEX", 8, 1), ("LIGHT_GEO_RAY", 8, 2),
            ("LIGHT_GEO_RESAMPLING", 1, 2), ("LIGHT_GEO_TREE_PREPASS", 8, 2), ("LIGHT_GEO_TREE_POSTPASS", 8, 2),
            ("LIGHT_GEO_BRIDGE_DISTANCE", 64, 1), ("LIGHT_GEO_BRIDGE_PHASE", 64, 1), ("LIGHT_GEO_BRIDGE_LIGHT_POINT", 8, 1),
            ("LIGHT_GEO_BRIDGE_VERTEX_COUNT", 8, 1), ("LIGHT_BSDF_CHOICE", 1, 1), ("LIGHT_BSDF_DIRECTION", 1, 1), ("LIGHT_BSDF_TRACE", 1, 1),
            ("LIGHT_BSDF_RR", 1, 1)]
    start, size, v = {}, {}, 0
    for name, count, sets in rows:
        start[name], size[name] = v, count
        v += count * sets + 1
    assert v == 577
    want = dict(start)
    want.update({"SUN_BSDF": start["LIGHT_SUN_BSDF"], "SUN_BSDF_METHOD": start["LIGHT_SUN_BSDF_METHOD"], "SUN_RAY": start["LIGHT_SUN_RAY"],
                 "SUN_RESAMPLING": start["LIGHT_SUN_RESAMPLING"], "BRIDGE_DISTANCE": start["LIGHT_GEO_BRIDGE_DISTANCE"], "BRIDGE_PHASE": start["LIGHT_GEO_BRIDGE_PHASE"],
                 "BRIDGE_LIGHT_POINT": start["LIGHT_GEO_BRIDGE_LIGHT_POINT"], "BRIDGE_VERTEX_COUNT": start["LIGHT_GEO_BRIDGE_VERTEX_COUNT"], "COUNT": 577,
                 "VOL_SUN_BSDF": start["LIGHT_SUN_BSDF"] + 1, "VOL_SUN_BSDF_METHOD": start["LIGHT_SUN_BSDF_METHOD"] + 1, "VOL_SUN_RAY": start["LIGHT_SUN_RAY"] + 1,
                 "VOL_SUN_RESAMPLING": start["LIGHT_SUN_RESAMPLING"] + 1, "VOL_GEO_RESAMPLING": start["LIGHT_GEO_RESAMPLING"] + 1,
                 "VOL_TREE_PREPASS": start["LIGHT_GEO_TREE_PREPASS"] + 8, "VOL_TREE_POSTPASS": start["LIGHT_GEO_TREE_POSTPASS"] + 8,
                 "VOL_GI_DIFFUSE": start["BSDF_DIFFUSE"], "VOL_GI_RESAMPLING": start["BSDF_RESAMPLING"], "VOL_AMBIENT_DIFFUSE": start["BSDF_DIFFUSE"] + 2,
                 "VOL_AMBIENT_RESAMPLING": start["BSDF_RESAMPLING"] + 2})
    text = "".join(open(os.path.join(ROOT, "oracle", f)).read() for f in ("o_rng.h", "o_sky.h", "o_volume.h"))
    found = {}
    for m in re.finditer(r"\bRT_([A-Z_0-9]+) *= *([0-9 +]+)[,\n]", text):
        found[m.group(1)] = sum(int(t) for t in m.group(2).split("+"))
    for m in re.finditer(r"#define (?:RT|RANDOM_TARGET)_([A-Z_0-9]+) +([0-9]+)u", text):
        found[m.group(1)] = int(m.group(2))
    assert len(found) >= 40
    for name, value in found.items():
        assert name in want, "constant without a row: " + name
        assert value == want[name], (name, value, want[name])
    # the device code repeats the numbers (dev_sampler.h, dev_sky.h, dev_volume.h)
    dev = "".join(open(os.path.join(ROOT, "luminary_amd", "csrc", "device", f)).read() for f in ("dev_sampler.h", "dev_sky.h", "dev_volume.h"))
    dev_names = {"kRndSkyStepOffset": "SKY_STEP_OFFSET", "kRndSkyInscatteringStep": "SKY_INSCATTERING_STEP", "kRndSunBsdf": "SUN_BSDF", "kRndSunResampling": "SUN_RESAMPLING",
                 "kRndVolumeIntersection": "VOLUME_INTERSECTION", "kRndSunInitialVertex": "LIGHT_SUN_INITIAL_VERTEX", "kRndGeoInitialVertex": "LIGHT_GEO_INITIAL_VERTEX",
                 "kRndVolSunBsdf": "VOL_SUN_BSDF", "kRndVolSunBsdfMethod": "VOL_SUN_BSDF_METHOD", "kRndVolSunRay": "VOL_SUN_RAY", "kRndVolSunResampling": "VOL_SUN_RESAMPLING",
                 "kRndVolGeoResampling": "VOL_GEO_RESAMPLING", "kRndVolTreePrepass": "VOL_TREE_PREPASS", "kRndVolTreePostpass": "VOL_TREE_POSTPASS",
                 "kRndBridgeDistance": "BRIDGE_DISTANCE", "kRndBridgePhase": "BRIDGE_PHASE", "kRndBridgeLightPoint": "BRIDGE_LIGHT_POINT",
                 "kRndBridgeVertexCount": "BRIDGE_VERTEX_COUNT", "kRndVolGiDiffuse": "VOL_GI_DIFFUSE", "kRndVolGiResampling": "VOL_GI_RESAMPLING",
                 "kRndVolAmbientDiffuse": "VOL_AMBIENT_DIFFUSE", "kRndVolAmbientResampling": "VOL_AMBIENT_RESAMPLING", "kRndLightTreePrepass": "LIGHT_GEO_TREE_PREPASS",
                 "kRndLightBsdfRR": "LIGHT_BSDF_RR", "kRndRussianRoulette": "RUSSIAN_ROULETTE"}
    for dev_name, row in dev_names.items():
        m = re.search(r"\b%s *= *([0-9]+)" % dev_name, dev)
        assert m, dev_name
        assert int(m.group(1)) == want[row], (dev_name, m.group(1), want[row])


def test_thin_fog_changes_nothing_and_thick_fog_dims_the_light(tmp_path):
    host = scenes.cornell_host(str(tmp_path), 40, 28, 4)
    base = oracle_lib.render(_view(host), 0, 32)[0].reshape(3, -1).mean(axis=1) / 32
    thin = oracle_lib.render(_view(_fogged(host, density=1e-4)), 0, 32)[0].reshape(3, -1).mean(axis=1) / 32
    assert np.allclose(thin, base, rtol=2e-3), (thin, base)
    fm, _, cnt = oracle_lib.render(_view(_fogged(host, density=300.0)), 0, 32)
    thick = fm.reshape(3, -1).mean(axis=1) / 32
    assert np.isfinite(fm).all()
    assert (thick < 0.7 * base).all() and (thick > 0.05 * base).all(), (thick, base)
    assert cnt[1] > 0


def test_fog_volume_path_matches_an_independent_intersection():
    """volume_compute_path (volume_utils.cuh:88-170) against a plain ray vs (vertical cylinder around the camera) x (slab below the fog height),
    written independently with numpy's quadratic formula."""
    rng = np.random.RandomState(5)
    n = 4000
    cam = np.array([3.0, 1.0, -2.0], dtype=np.float32)
    origins = (rng.uniform(-60, 60, (n, 3)) * [1, 0.6, 1]).astype(np.float32) + cam
    dirs = rng.normal(size=(n, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dirs = dirs[np.abs(dirs[:, 1]) > 0.02][: n // 2].astype(np.float32)   # the near-horizontal special case is the reference's own approximation
    origins = origins[: len(dirs)]
    limits = rng.uniform(1.0, 200.0, len(dirs)).astype(np.float32)
    height, dist = 12.0, 40.0
    out = np.zeros((len(dirs), 2), dtype=np.float32)
    L.oracle_probe_volume_path((C.c_float * 3)(*cam), C.c_float(dist), C.c_float(height), C.c_uint32(len(dirs)), origins.ctypes.data_as(C.c_void_p),
                               dirs.ctypes.data_as(C.c_void_p), limits.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    o, d = origins.astype(np.float64), dirs.astype(np.float64)
    # slab y in (-inf, height]
    t_top = (height - o[:, 1]) / d[:, 1]
    lo_y = np.where(d[:, 1] > 0, 0.0, np.maximum(t_top, 0.0))
    hi_y = np.where(d[:, 1] > 0, t_top, np.inf)
    # cylinder
    ox, oz = o[:, 0] - cam[0], o[:, 2] - cam[2]
    a = d[:, 0] ** 2 + d[:, 2] ** 2
    b = 2 * (ox * d[:, 0] + oz * d[:, 2])
    c = ox ** 2 + oz ** 2 - dist ** 2
    disc = b * b - 4 * a * c
    ok = disc >= 0
    sq = np.sqrt(np.where(ok, disc, 0))
    t0, t1 = (-b - sq) / (2 * a), (-b + sq) / (2 * a)
    # The reference measures the cylinder's two distances in the horizontal plane (it normalises the direction's x and z, volume_utils.cuh:134-157) and
    # mixes them with the slab's distances along the ray; kept (reference quirks): a steep ray leaves the disk later than its fog ends.
    t0, t1 = t0 * np.sqrt(a), t1 * np.sqrt(a)
    lo = np.maximum(np.maximum(t0, 0.0), lo_y)
    hi = np.minimum(np.minimum(t1, hi_y), limits)
    length = np.where(ok & (hi > lo), hi - lo, 0.0)
    got_len = np.where(out[:, 0] >= 0, out[:, 1], 0.0)
    assert np.allclose(got_len, length, rtol=2e-3, atol=2e-3), np.abs(got_len - length).max()
    inside = length > 1e-2
    assert np.allclose(out[inside, 0], lo[inside], rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("diameter", [10.0, 3.0, 0.7])
def test_fog_phase_function_is_normalised_and_its_sampler_follows_it(tmp_path, diameter):
    """Jendersie-Eon phase function (math.cuh:1169-1247): integrates to 1 over the sphere; the sampler (math.cuh:1274-1323) draws cos(theta) with that density.
    The diameters cover three branches of the parameter fit."""
    view = _view(_fogged(scenes.cornell_host(str(tmp_path), 8, 8, 1), diameter=diameter))
    cos = np.linspace(-1, 1, 400001)
    cos32 = cos.astype(np.float32)
    val = np.zeros_like(cos32)
    L.oracle_probe_fog_phase(C.byref(view), C.c_uint32(len(cos32)), cos32.ctypes.data_as(C.c_void_p), val.ctypes.data_as(C.c_void_p))
    integral = 2 * np.pi * np.trapezoid(val.astype(np.float64), cos)
    assert abs(integral - 1.0) < 3e-3, integral
    n = 200000
    rnd = np.random.RandomState(3).rand(n, 3).astype(np.float32)
    got = np.zeros(n, dtype=np.float32)
    L.oracle_probe_fog_phase_sample(C.byref(view), C.c_uint32(n), rnd.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p))
    assert np.isfinite(got).all() and (np.abs(got) <= 1.0 + 1e-4).all()
    cdf = np.concatenate([[0.0], np.cumsum(0.5 * (val[1:] + val[:-1]).astype(np.float64) * np.diff(cos))]) * 2 * np.pi
    for q in (-0.5, 0.0, 0.5, 0.9, 0.99):
        want = np.interp(q, cos, cdf)
        have = (got <= q).mean()
        assert abs(have - want) < 5e-3, (diameter, q, have, want)


def test_distance_sampling_is_consistent_with_its_density():
    """volume_sample_intersection_bounded and its pdf (volume_utils.cuh:204-214): the estimator of the scattered fraction, sigma * T(t) / pdf(t), is exactly
    1 - exp(-sigma * max) for every sample (perfect importance sampling), and the samples are exponentially distributed."""
    sigma, max_len = 0.05, 30.0
    rnd = np.random.RandomState(9).rand(50000).astype(np.float32)
    t = np.zeros_like(rnd); pdf = np.zeros_like(rnd)
    L.oracle_probe_volume_sampling(C.c_float(sigma), C.c_float(max_len), C.c_uint32(len(rnd)), rnd.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p),
                                   pdf.ctypes.data_as(C.c_void_p))
    assert (t >= 0).all() and (t <= max_len * 1.0001).all()
    est = sigma * np.exp(-sigma * t.astype(np.float64)) / pdf
    assert np.allclose(est, 1 - np.exp(-sigma * max_len), rtol=1e-4)
    for q in (5.0, 15.0, 25.0):
        want = (1 - np.exp(-sigma * q)) / (1 - np.exp(-sigma * max_len))
        assert abs((t <= q).mean() - want) < 8e-3


def test_one_vertex_bridges_match_the_quadrature_of_their_estimator(tmp_path):
    """Bridges limited to one vertex connect a point of the camera ray straight to a point of the light. A camera ray through fog past a small emissive
    triangle, nothing else in the scene, black sky: the oracle's estimate against the double integral (ray x triangle, numpy) of what light_bridges.cuh
    evaluates: Le * [sigma T(t)] * HG_0.85(cos) * [sigma T(r)] / r^2. This pins the densities the sampler divides by (the two-interval sampling of the
    initial vertex, area sampling of the light, r^2 of the one-segment path, the resampling of the eight candidates) - every one of them integrates out.
    Two things in that integrand are the reference's own and kept: the scattering coefficient appears twice (light_bridges.cuh:207-213 on top of :257-259) and
    the emitter's cosine does not appear, so this is not physical single scattering; DESIGN.md lists both."""
    density, le = 60.0, 40.0
    sigma = 0.001 * density
    tri = np.array([[2.0, 3.0, -9.0], [2.0, 3.0, -11.0], [4.0, 3.0, -9.0]])
    host = scenes.probe_light_scene(str(tmp_path), tri, (le, le, le), width=8, height=8, bounces=0)
    _fogged(host, density=density, height=1000.0, dist=1000.0)
    st = host.get_settings(); st.bridge_max_num_vertices = 1; host.set_settings(st)
    view = _view(host)
    spp = 30000
    px = np.array([4 + 4 * 8], dtype=np.uint32)
    fm, _, cnt = oracle_lib.render(view, 0, spp, pixels=px)
    got = fm.reshape(3, -1)[:, 0] / spp
    assert cnt[1] > 0.2 * spp, "bridge segments were traced"
    # the pixel's footprint: the rays of its first 48 samples (the estimate changes by 10 % from one pixel row to the next)
    rays = []
    for sample in range(48):
        ray = np.zeros(6, dtype=np.float32)
        L.oracle_camera_ray(C.byref(view), C.c_uint32(4), C.c_uint32(4), C.c_uint32(sample), ray.ctypes.data_as(C.c_void_p))
        rays.append(ray.astype(np.float64))
    rng = np.random.RandomState(1)
    u = rng.rand(1200, 2); su = np.sqrt(u[:, 0])
    pts = tri[0] + (tri[1] - tri[0]) * (su * (1 - u[:, 1]))[:, None] + (tri[2] - tri[0]) * (su * u[:, 1])[:, None]
    area = 0.5 * np.linalg.norm(np.cross(tri[1] - tri[0], tri[2] - tri[0]))
    ts = (np.arange(2000) + 0.5) * (80.0 / 2000)
    g = 0.85
    totals = []
    for ray in rays:
        o, d = ray[:3], ray[3:]
        x = o[None, :] + ts[:, None] * d[None, :]                   # [t, 3]
        w = pts[None, :, :] - x[:, None, :]                         # [t, p, 3]
        r = np.linalg.norm(w, axis=2)
        c = (w @ d) / r
        hg = (1 - g * g) / (4 * np.pi * (1 + g * g - 2 * g * c) ** 1.5)
        totals.append((sigma * np.exp(-sigma * ts)[:, None] * hg * sigma * np.exp(-sigma * r) / r ** 2).mean(axis=1).sum())
    want = le * area * np.mean(totals) * (ts[1] - ts[0])
    assert want > 1e-6
    assert abs(got[0] - want) < 0.02 * want, (got, want)
    assert got[0] == got[1] == got[2]


# ---------------------------------------------------------------- GPU: HIP == oracle

def _parity(host, samples=3, first=0, spp_pass=2):
    from luminary_amd.core import Core
    view = _view(host)
    core = Core(0)
    try:
        assert core.flavour == "exact"
        core.upload(view)
        core.set_pixels(None)
        core.reset_counters()
        core.render(first, samples, samples_per_pass=spp_pass)
        fm, sm = core.accumulators()
        ofm, osm, ocnt = oracle_lib.render(view, first, samples)
        assert np.isfinite(ofm).all()
        assert np.array_equal(fm, ofm), "first moment: %d of %d differ, max %g" % ((fm != ofm).sum(), fm.size, np.abs(fm - ofm).max())
        assert np.array_equal(sm, osm)
        assert core.query_counters()[:4] == [int(x) for x in ocnt[:4]], (core.query_counters()[:4], list(ocnt[:4]))
        return ofm, ocnt
    finally:
        core.close()


@pytest.mark.gpu
@pytest.mark.parametrize("density", [5.0, 150.0])
def test_fogged_cornell_matches_the_oracle(tmp_path, density):
    """Cornell box with its ceiling light in fog: bridges on the camera segment, transmittance on every light sample, scattering events and bounces."""
    host = _fogged(scenes.cornell_host(str(tmp_path), 48, 32, 4), density=density)
    _, cnt = _parity(host, samples=4)
    assert cnt[1] > 48 * 32  # bridge segments are counted with the shadow rays


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [SKY_MODE_CONSTANT_COLOR, SKY_MODE_DEFAULT, SKY_MODE_HDRI])
def test_fogged_zoo_matches_the_oracle(mode):
    """The material zoo (320 emissive triangles: the light tree is descended for the volume context too) in all three sky modes: ambient and sun scattered in
    by the fog, the sky fast path of the scattering-event kernel, a low fog ceiling that the camera looks out of."""
    host = scenes.zoo_scene(64, 40, 5, sky_mode=mode)
    _fogged(host, density=25.0, height=6.0, dist=60.0, diameter=4.0)
    _parity(host, samples=3)


@pytest.mark.gpu
def test_fog_with_aerial_perspective_and_depth_limits(tmp_path):
    """Fog + aerial perspective (the in-scattering kernel of the sky runs on the events' new depths, also for paths the fast path ended), and the depth limits:
    0 (no bounce kernel at all) and 1."""
    host = scenes.zoo_scene(56, 36, 3, sky_mode=SKY_MODE_HDRI)
    sky = host.get_sky(); sky.aerial_perspective = True; host.set_sky(sky)
    _fogged(host, density=30.0, height=40.0, dist=80.0)
    _parity(host, samples=2)
    for depth in (0, 1):
        h = _fogged(scenes.cornell_host(str(tmp_path / ("d%d" % depth)), 40, 28, depth), density=80.0)
        _parity(h, samples=2)


@pytest.mark.gpu
def test_few_bridge_vertices_and_a_scene_without_lights(tmp_path):
    host = _fogged(scenes.cornell_host(str(tmp_path), 40, 28, 3), density=120.0)
    st = host.get_settings(); st.bridge_max_num_vertices = 2; host.set_settings(st)
    _parity(host, samples=3)
    # no emissive triangle: no light tree, no bridges, only the sky lights the fog
    dark = scenes.edge_scene("no_lights", 40, 28, 3)
    _fogged(dark, density=60.0)
    _parity(dark, samples=3)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2, 4])
def test_debug_shading_modes_see_the_fog_events(mode):
    """The debug queue keeps the scattering-event kernel (device_renderer.c:145-147): rays that leave the scene show the sky through the fog (fast path),
    paths that scatter stay black, surface hits keep their debug colour."""
    from luminary_amd.core import Core
    host = _fogged(scenes.zoo_scene(64, 40, 4), density=40.0, height=30.0, dist=50.0)
    st = host.get_settings(); st.shading_mode = mode; host.set_settings(st)
    view = _view(host)
    core = Core(0)
    try:
        core.upload(view)
        core.set_pixels(None)
        core.render(0, 3, samples_per_pass=3)
        fm, sm = core.accumulators()
        ofm, osm, _ = oracle_lib.render(view, 0, 3)
        assert np.array_equal(fm, ofm) and np.array_equal(sm, osm)
        plain = scenes.zoo_scene(64, 40, 4)
        st = plain.get_settings(); st.shading_mode = mode; plain.set_settings(st)
        assert not np.array_equal(ofm, oracle_lib.render(_view(plain), 0, 3)[0]), "the events change the debug image"
    finally:
        core.close()


@pytest.mark.gpu
def test_fast_flavour_renders_the_same_fog(tmp_path):
    """The default (fast) flavour on the fogged Cornell box at 256 spp against the exact one: same estimator, different rounding."""
    from luminary_amd.core import Core
    host = _fogged(scenes.cornell_host(str(tmp_path), 48, 32, 4), density=60.0)
    view = _view(host)
    frames = {}
    for flavour in ("exact", "fast"):
        core = Core(0)
        try:
            core.set_flavour(flavour)
            core.upload(view)
            core.set_pixels(None)
            core.render(0, 256, samples_per_pass=8)
            frames[flavour] = core.accumulators()[0].astype(np.float64) / 256
        finally:
            core.close()
    a, b = frames["exact"], frames["fast"]
    assert np.isfinite(b).all()
    rel_l2 = np.linalg.norm(a - b) / np.linalg.norm(a)
    assert rel_l2 < 0.05, rel_l2                      # two 256-spp estimates with decorrelating roundings
    assert abs(b.sum() / a.sum() - 1.0) < 5e-3        # no bias in the frame sum


@pytest.mark.gpu
def test_tile_partition_and_adaptive_pass_shapes_with_fog_and_particles(tmp_path):
    """What the multi-GPU path and the batching rely on, with the volume kernels in the loop: a 3-way tile partition of the frame reproduces the full frame
    bit for bit (every path owns its in-scattering records and its 17 visibility slots), and so do two sample ids in one pass vs two passes."""
    from luminary_amd.core import Core
    from luminary_amd.distributed import tile_pixels
    host = _fogged(scenes.cornell_host(str(tmp_path), 96, 64, 4), density=70.0)
    p = host.get_particles(); p.active, p.count, p.size, p.scale = True, 600, 5.0, 1.5; host.set_particles(p)
    view = _view(host)
    core = Core(0)
    try:
        core.upload(view)
        core.set_pixels(None)
        core.render(0, 2, samples_per_pass=2)
        full, full_sm = core.accumulators()
        ofm, osm, _ = oracle_lib.render(view, 0, 2)
        assert np.array_equal(full, ofm) and np.array_equal(full_sm, osm)
        core.set_pixels(None)  # resets the accumulators
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
def test_fog_and_particles_through_the_host_api(tmp_path):
    """luminary_host_set_fog / set_particles, then the library's own render entry (luminary_ext_render_samples): its accumulators equal the oracle's."""
    host = scenes.cornell_host(str(tmp_path), 48, 32, 3)
    _fogged(host, density=80.0)
    p = host.get_particles(); p.active, p.count, p.size, p.scale = True, 400, 6.0, 1.5; host.set_particles(p)
    assert host.get_fog().active and host.get_particles().count == 400
    host.render_samples(0, 2)
    fm, sm = host.accumulators()
    ofm, osm, _ = oracle_lib.render(_view(host), 0, 2)
    assert np.array_equal(fm, ofm) and np.array_equal(sm, osm)
    plain = scenes.cornell_host(str(tmp_path / "plain"), 48, 32, 3)
    assert not np.array_equal(ofm, oracle_lib.render(_view(plain), 0, 2)[0])
