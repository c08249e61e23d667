"""Procedural sky (first part): look-up tables, the colour of rays that leave the scene, the sun disk.
CPU part: properties of the oracle restatement. GPU part: tables and images against the oracle, bit for bit."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from luminary_amd import SKY_MODE_DEFAULT, SKY_MODE_HDRI, scenes
from luminary_amd.core import Core

W, H = 72, 40


def _scene(bounces=3, altitude=0.5, azimuth=3.141):
    host = scenes.example_scene(W, H, bounces, sphere_segments=8, ground_res=16, num_objects=24, num_lights=4)
    sky = host.get_sky()
    sky.mode = SKY_MODE_DEFAULT
    sky.altitude, sky.azimuth = altitude, azimuth
    host.set_sky(sky)
    return host


_oracle_luts = oracle_lib.sky_luts
_with_sky_luts = oracle_lib.with_sky_luts


def _sky_color(v, origin, ray, include_sun, offset=0.5):
    out = (C.c_float * 3)()
    oracle_lib.lib().oracle_sky_color(C.byref(v), (C.c_float * 3)(*origin), (C.c_float * 3)(*[float(x) for x in ray]), C.c_int(1 if include_sun else 0),
                                      C.c_float(offset), out)
    return np.array(list(out), dtype=np.float32)


def test_oracle_sky_tables_and_colours():
    host = _scene()
    v = _with_sky_luts(host.device_scene())
    tm, ms = v._sky_keep
    tm4 = tm.reshape(2, 64, 256, 4)
    assert np.isfinite(tm).all() and tm.min() >= 0.0 and tm.max() <= 1.0
    assert np.isfinite(ms).all() and ms.min() >= 0.0 and ms.max() > 0.0
    # looking straight up from higher in the atmosphere there is less air above: transmittance grows with height (u = 0 is "up")
    assert (np.diff(tm4[0, :, 0, 1]) >= -1e-6).all() and tm4[0, -1, 0, 1] > tm4[0, 0, 0, 1]
    # the host layer's sun position: distance to the earth's centre is the sun's distance, direction from azimuth / altitude
    sun = np.array(list(v.sky_sun_pos), dtype=np.float64) + np.array([0.0, 6371.0 + 0.1, 0.0])
    assert abs(np.linalg.norm(sun) - 149597870.0) < 1e3
    assert abs(sun[1] / np.linalg.norm(sun) - np.sin(0.5)) < 1e-5
    zenith = _sky_color(v, (0, 1, 0), (0, 1, 0), False)
    horizon = _sky_color(v, (0, 1, 0), (0.9998, 0.02, 0), False)
    assert zenith[2] > zenith[0] * 2 and zenith.min() > 0.0, "blue sky overhead"
    assert horizon.sum() > zenith.sum(), "brighter towards the horizon"
    sun_dir = np.array(list(v.sky_sun_pos)) / np.linalg.norm(list(v.sky_sun_pos))
    with_disk, without = _sky_color(v, (0, 1, 0), sun_dir, True), _sky_color(v, (0, 1, 0), sun_dir, False)
    assert with_disk.min() > 1000.0 * without.max(), "the sun disk is only added for camera / emission-allowed rays"
    down = _sky_color(v, (0, 1, 0), (0, -1, 0), True)
    assert down.sum() < 0.05 * horizon.sum(), "below the camera is the planet: only the hundred metres of air above the ground scatter"
    # sunset: the sky near the sun turns red
    low = _with_sky_luts(_scene(altitude=0.02).device_scene())
    sd = np.array(list(low.sky_sun_pos)) / np.linalg.norm(list(low.sky_sun_pos))
    glow = _sky_color(low, (0, 1, 0), (sd[0], sd[1] + 0.05, sd[2]) / np.linalg.norm((sd[0], sd[1] + 0.05, sd[2])), False)
    assert glow[0] > glow[2]


@pytest.mark.gpu
def test_sky_tables_match_the_oracle():
    host = _scene()
    view = host.device_scene()
    tm, ms = _oracle_luts(view)
    core = Core(0)
    try:
        core.upload(oracle_lib.with_luts(view))  # sky LUT pointers NULL -> generated on the GPU
        got_tm, got_ms = core.download_sky_luts()
        assert np.array_equal(got_tm, tm), "transmittance table: %d of %d differ" % ((got_tm != tm).sum(), tm.size)
        assert np.array_equal(got_ms, ms), "multiscattering table: %d of %d differ" % ((got_ms != ms).sum(), ms.size)
    finally:
        core.close()


@pytest.mark.gpu
@pytest.mark.parametrize("altitude", [0.5, 0.03])
def test_render_parity_with_the_procedural_sky(altitude):
    """Open scene under the atmosphere: camera rays see the sky and the sun disk, bounce rays gather sky light."""
    host = _scene(altitude=altitude)
    view = _with_sky_luts(host.device_scene())
    core = Core(0)
    try:
        core.upload(view)
        core.set_pixels(None)
        core.reset_counters()
        core.render(0, 3, samples_per_pass=2)
        fm, sm = core.accumulators()
        ofm, osm, ocnt = oracle_lib.render(view, 0, 3)
        assert np.array_equal(fm, ofm), "first moment: %d of %d differ, max %g" % ((fm != ofm).sum(), fm.size, np.abs(fm - ofm).max())
        assert np.array_equal(sm, osm)
        assert core.query_counters()[:4] == [int(x) for x in ocnt[:4]]
        assert fm.reshape(3, H, W)[:, :4].mean() > 0.0, "the top rows see the sky"
    finally:
        core.close()


def test_oracle_sun_lights_the_ground():
    """Sun next-event estimation: with the sun up the ground is lit by it through extra visibility rays; below the horizon no sun ray is
    traced and only the faint sky remains."""
    day = _with_sky_luts(_scene(altitude=0.5).device_scene())
    night = _with_sky_luts(_scene(altitude=-0.2).device_scene())
    fm_d, _, cnt_d = oracle_lib.render(day, 0, 1)
    fm_n, _, cnt_n = oracle_lib.render(night, 0, 1)
    assert cnt_d[0] == cnt_n[0] and cnt_d[3] == cnt_n[3], "same paths (the sun does not change the bounce directions)"
    assert cnt_d[1] > cnt_n[1] + 1000, "sun rays are counted with the shadow rays"
    ground_d, ground_n = fm_d.reshape(3, H, W)[:, -10:].mean(), fm_n.reshape(3, H, W)[:, -10:].mean()
    assert ground_d > 20.0 * ground_n


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["no_ozone", "dense_high_camera", "few_steps_translucent"])
def test_sky_variants_match_the_oracle(variant):
    """Other corners of the atmosphere's parameter space: ozone off, a denser atmosphere seen from 3 km up with a shifted planet centre,
    very few ray-march steps; the zoo scene adds translucent and transparent surfaces to the sun sampling."""
    if variant == "few_steps_translucent":
        host = scenes.zoo_scene(64, 40, 5, sky_mode=SKY_MODE_DEFAULT)
    else:
        host = _scene(bounces=2)
    sky = host.get_sky()
    sky.mode = SKY_MODE_DEFAULT
    if variant == "no_ozone":
        sky.ozone_absorption = False
        sky.mie_diameter = 20.0            # the first branch of the phase-function fit
    elif variant == "dense_high_camera":
        sky.base_density, sky.rayleigh_density, sky.mie_density, sky.ground_visibility = 1.5, 1.2, 2.0, 20.0
        sky.geometry_offset.x, sky.geometry_offset.y, sky.geometry_offset.z = 5.0, 3.0, -2.0
        sky.sun_strength, sky.multiscattering_factor, sky.mie_diameter = 2.0, 0.5, 0.5
        sky.altitude, sky.azimuth = 0.15, 1.0
    else:
        sky.steps = 3
        sky.altitude = 1.2
    host.set_sky(sky)
    view = oracle_lib.with_sky_luts(host.device_scene())
    core = Core(0)
    try:
        core.upload(oracle_lib.with_luts(host.device_scene()))  # tables generated on the GPU
        got_tm, got_ms = core.download_sky_luts()
        tm, ms = oracle_lib.sky_luts(host.device_scene())
        assert np.array_equal(got_tm, tm) and np.array_equal(got_ms, ms), "tables"
        core.set_pixels(None)
        core.reset_counters()
        core.render(1, 2, samples_per_pass=2)
        fm, sm = core.accumulators()
        ofm, osm, ocnt = oracle_lib.render(view, 1, 2)
        assert np.array_equal(fm, ofm), "first moment: %d of %d differ, max %g" % ((fm != ofm).sum(), fm.size, np.abs(fm - ofm).max())
        assert np.array_equal(sm, osm)
        assert core.query_counters()[:4] == [int(x) for x in ocnt[:4]]
    finally:
        core.close()


def _night_scene(fov=0.03):
    """The example scene at night through a long lens: sun below the horizon, the moon placed on the optical axis, stars around it."""
    host = _scene(bounces=2, altitude=-0.4, azimuth=1.0)
    scenes.set_camera(host, (0.0, 6.0, 28.0), (0.25, 0.3, 0.0), fov=fov)
    view = host.device_scene()
    ray = (C.c_float * 6)()
    oracle_lib.lib().oracle_camera_ray(C.byref(view), C.c_uint32(W // 2), C.c_uint32(H // 2), C.c_uint32(0), ray)
    d = np.array(list(ray)[3:], dtype=np.float64)
    sky = host.get_sky()
    # the moon's position is direction * 384399 km - (0, R, 0) - offset in sky space (device_structs.c:152-165) while the camera sits at
    # (0, R, 0) + offset + position / 1000: solve for the direction whose moon lies on the optical axis as seen from the camera
    off = np.array([sky.geometry_offset.x, sky.geometry_offset.y, sky.geometry_offset.z], dtype=np.float64)
    base = np.array([0.0, 6.0, 28.0]) * 0.001 + np.array([0.0, 6371.0, 0.0]) + off + np.array([0.0, 6371.0, 0.0]) + off
    b, c = 2.0 * np.dot(base, d), np.dot(base, base) - 384399.0 ** 2
    t = (-b + np.sqrt(b * b - 4.0 * c)) / 2.0
    q = (base + t * d) / 384399.0
    sky.moon_altitude, sky.moon_azimuth = float(np.arcsin(q[1])), float(np.arctan2(q[2], q[0]))
    sky.stars_intensity = 50.0
    host.set_sky(sky)
    return host, d


def test_oracle_moon_and_stars():
    host, d = _night_scene()
    view = host.device_scene()
    assert view.sky_moon_albedo_tex == view.num_textures - 2 and view.sky_moon_normal_tex == view.num_textures - 1
    assert view.sky_stars_count == 10000
    offsets = np.ctypeslib.as_array(C.cast(view.sky_stars_offsets, C.POINTER(C.c_uint32)), shape=(64 * 32 + 1,))
    stars = np.ctypeslib.as_array(C.cast(view.sky_stars, C.POINTER(C.c_float)), shape=(10000, 4))
    assert offsets[0] == 0 and offsets[-1] == 10000 and (np.diff(offsets.astype(np.int64)) >= 0).all()
    cell = (stars[:, 1] * np.float32(10.0)).astype(np.uint32) + ((stars[:, 0] + np.float32(3.141592653589) * np.float32(0.5)) * np.float32(10.0)).astype(np.uint32) * 64
    assert (np.diff(cell.astype(np.int64)) >= 0).all(), "stars are stored cell by cell"
    v = _with_sky_luts(view)
    moon = _sky_color(v, (0.0, 6.0, 28.0), d, True)
    off = _sky_color(v, (0.0, 6.0, 28.0), (d + np.array([0.0, 0.02, 0.0])) / np.linalg.norm(d + np.array([0.0, 0.02, 0.0])), True)
    assert moon.min() > 0.0 and moon.sum() > 50.0 * off.sum(), "the moon's lit face against the night sky"
    assert np.array_equal(_sky_color(v, (0.0, 6.0, 28.0), d, False), _sky_color(v, (0.0, 6.0, 28.0), d, False)) and _sky_color(v, (0.0, 6.0, 28.0), d, False).sum() < moon.sum() / 50.0
    # a ray straight at a star picks it up
    k = int(np.argmax(stars[:, 3] * (stars[:, 0] > 0.3)))
    alt, az = float(stars[k, 0]), float(stars[k, 1])
    sd = np.array([np.cos(az) * np.cos(alt), np.sin(alt), np.sin(az) * np.cos(alt)])
    lit = _sky_color(v, (0.0, 6.0, 28.0), sd, True)
    dark = _sky_color(v, (0.0, 6.0, 28.0), sd, False)
    assert lit.sum() > dark.sum() + 1e-4


@pytest.mark.gpu
def test_night_sky_matches_the_oracle():
    """Moon (textured, lit by the sun from below the horizon) and stars through a long lens, bit for bit."""
    host, _ = _night_scene()
    view = _with_sky_luts(host.device_scene())
    core = Core(0)
    try:
        core.upload(view)
        core.set_pixels(None)
        core.render(0, 2, samples_per_pass=2)
        fm, sm = core.accumulators()
        ofm, osm, _ = oracle_lib.render(view, 0, 2)
        assert np.array_equal(fm, ofm), "first moment: %d of %d differ, max %g" % ((fm != ofm).sum(), fm.size, np.abs(fm - ofm).max())
        assert np.array_equal(sm, osm)
        img = fm.reshape(3, H, W).sum(axis=0)
        assert img[H // 2 - 3:H // 2 + 3, W // 2 - 3:W // 2 + 3].mean() > 20.0 * np.median(img), "the moon is in the middle of the frame"
    finally:
        core.close()


def _oracle_hdri(view, origin, dim, samples):
    out = np.zeros((dim, dim, 4), dtype=np.float32)
    oracle_lib.lib().oracle_sky_hdri(C.byref(view), (C.c_float * 3)(*origin), C.c_uint32(dim), C.c_uint32(samples), out.ctypes.data_as(C.c_void_p))
    return out


def test_oracle_hdri_bake_is_a_sky_panorama():
    v = _with_sky_luts(_scene().device_scene())
    img = _oracle_hdri(v, (0.0, 6.0, 28.0), 16, 3)
    assert np.isfinite(img).all() and (img[..., 3] == 1.0).all(), "fourth channel: the clouds' transmittance, 1 without clouds"
    assert img[:8, :, :3].mean() > 20.0 * img[9:, :, :3].mean(), "rows above the horizon hold the sky, rows below it the thin air above the ground"
    assert img[1:7, :, 2].mean() > img[1:7, :, 0].mean(), "blue overhead"


@pytest.mark.gpu
@pytest.mark.parametrize("dim,samples", [(24, 40), (9, 5)])
def test_hdri_bake_matches_the_oracle(dim, samples):
    """lumc_sky_hdri_build: jittered samples per texel shared by 32 lanes, their means through the trimmed mean (more samples than
    lanes, and fewer buckets than lanes)."""
    host = _scene(altitude=0.3)
    view = _with_sky_luts(host.device_scene())
    core = Core(0)
    try:
        core.upload(view)
        got = core.sky_hdri_build((1.0, 6.0, 28.0), dim, samples)
        want = _oracle_hdri(view, (1.0, 6.0, 28.0), dim, samples)
        assert np.array_equal(got, want), "%d of %d values differ, max %g" % ((got != want).sum(), got.size, np.abs(got - want).max())
    finally:
        core.close()


# ---- sky mode HDRI: the baked panorama is the sky; the sun is sampled beside it ----
def _hdri_scene(dim=32, samples=3, altitude=0.4, bounces=3):
    host = _scene(bounces=bounces, altitude=altitude)
    sky = host.get_sky()
    sky.mode = SKY_MODE_HDRI
    sky.hdri_dim, sky.hdri_samples = dim, samples
    host.set_sky(sky)
    return host


def test_oracle_hdri_mode():
    host = _hdri_scene()
    plain = host.device_scene()
    assert plain.sky_mode == SKY_MODE_HDRI and plain.sky_hdri_dim == 32 and plain.sky_hdri_samples == 3 and not plain.sky_hdri
    cam = host.get_camera()
    assert tuple(plain.sky_hdri_origin) == (cam.pos.x, cam.pos.y, cam.pos.z), "the panorama is baked from the camera position"
    view = oracle_lib.with_sky_hdri(plain)
    fm, sm, cnt = oracle_lib.render(view, 0, 2)
    img = fm.reshape(3, H, W)
    assert np.isfinite(fm).all() and img[:, :4].mean() > 0.0, "the top rows see the panorama"
    # a black panorama leaves the sun (disk and sampled light); a panorama of ones adds ambient light everywhere
    black = oracle_lib.with_sky_hdri(plain, np.zeros((4, 4, 4), np.float32))
    white = oracle_lib.with_sky_hdri(plain, np.ones((4, 4, 4), np.float32))
    fb, _, cb = oracle_lib.render(black, 0, 2)
    fw, _, cw = oracle_lib.render(white, 0, 2)
    assert fb.sum() > 0.0, "the sun still lights the ground"
    assert (fw >= fb).all() and fw.sum() > fb.sum()
    assert cw[1] > cb[1], "ambient samples are extra visibility rays"
    # the lookup: texel (x, y) = floor of the equirectangular coordinate, v = 0 at the zenith
    pano = np.zeros((8, 8, 4), np.float32)
    pano[0] = (0.0, 5.0, 0.0, 0.0)   # the row around the zenith is green
    pano[7] = (5.0, 0.0, 0.0, 0.0)   # the row around the nadir is red
    v2 = oracle_lib.with_sky_hdri(plain, pano)
    out = (C.c_float * 3)()
    lib = oracle_lib.lib()
    lib.oracle_sky_hdri_color(C.byref(v2), (C.c_float * 3)(0, 1, 0), (C.c_float * 3)(0.0, 1.0, 0.0), C.c_uint32(0), out)
    assert tuple(out) == (0.0, 5.0, 0.0)
    lib.oracle_sky_hdri_color(C.byref(v2), (C.c_float * 3)(0, 1, 0), (C.c_float * 3)(0.0, -0.999, 0.0447), C.c_uint32(0), out)
    assert tuple(out) == (5.0, 0.0, 0.0)


def test_hdri_origin_follows_the_camera_only_when_the_sky_is_dirty():
    """sky_hdri_update (device_sky.c:249-281) runs on sky changes and on luminary_host_request_sky_hdri_build, not on camera moves."""
    host = _hdri_scene()
    first = tuple(host.device_scene().sky_hdri_origin)
    cam = host.get_camera()
    cam.pos.x += 3.0
    host.set_camera(cam)
    assert tuple(host.device_scene().sky_hdri_origin) == first
    host.request_sky_hdri_build()
    assert tuple(host.device_scene().sky_hdri_origin) == (cam.pos.x, cam.pos.y, cam.pos.z)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["baked_at_upload", "given_panorama", "low_sun"])
def test_render_parity_in_hdri_mode(case):
    host = _hdri_scene(dim=24, samples=2, altitude=0.05 if case == "low_sun" else 0.4)
    plain = host.device_scene()
    if case == "given_panorama":
        rng = np.random.default_rng(5)
        pano = rng.random((7, 7, 4), dtype=np.float32) * 3.0
        oracle_view = oracle_lib.with_sky_hdri(plain, pano)
        gpu_view = oracle_view
    else:
        oracle_view = oracle_lib.with_sky_hdri(plain)   # the oracle's bake
        gpu_view = oracle_lib.with_luts(plain)           # no tables, no panorama: both are made on the GPU at upload
    core = Core(0)
    try:
        core.upload(gpu_view)
        if case != "given_panorama":
            got = core.sky_hdri_download()
            assert np.array_equal(got, oracle_view._hdri_keep), "the bake at upload"
        core.set_pixels(None)
        core.reset_counters()
        core.render(0, 3, samples_per_pass=2)
        fm, sm = core.accumulators()
        ofm, osm, ocnt = oracle_lib.render(oracle_view, 0, 3)
        assert np.array_equal(fm, ofm), "first moment: %d of %d differ, max %g" % ((fm != ofm).sum(), fm.size, np.abs(fm - ofm).max())
        assert np.array_equal(sm, osm)
        assert core.query_counters()[:4] == [int(x) for x in ocnt[:4]]
    finally:
        core.close()


@pytest.mark.gpu
def test_hdri_rebuild_on_request_through_the_host_api():
    """luminary_host_request_sky_hdri_build: the next render uses a panorama baked from the camera's new position."""
    host = _hdri_scene(dim=16, samples=2)
    host.render(1)
    a = host.core_sky_hdri()
    cam = host.get_camera()
    cam.pos.y += 2000.0   # two kilometres up: a visibly different sky
    host.set_camera(cam)
    host.render(1)
    assert np.array_equal(host.core_sky_hdri(), a), "moving the camera alone keeps the panorama"
    host.request_sky_hdri_build()
    host.render(1)
    b = host.core_sky_hdri()
    assert not np.array_equal(a, b)
    want = oracle_lib.sky_hdri(oracle_lib.with_sky_luts(host.device_scene()))
    assert np.array_equal(b, want)


# ---- aerial perspective: the air between a ray's origin and its hit (sky.aerial_perspective) ----
def _distant_scene(aerial, mode=SKY_MODE_DEFAULT, scale=400.0):
    """The example scene blown up to kilometres, so that there is enough air in front of the geometry to see."""
    host = scenes.example_scene(W, H, 3, sphere_segments=8, ground_res=16, num_objects=24, num_lights=4)
    sky = host.get_sky()
    sky.mode = mode
    sky.altitude, sky.azimuth = 0.6, 2.0
    sky.aerial_perspective = aerial
    sky.hdri_dim, sky.hdri_samples = 16, 2
    host.set_sky(sky)
    n = host.counts()[2]
    for i in range(n):
        inst = host.get_instance(i)
        inst.position.x, inst.position.y, inst.position.z = inst.position.x * scale, inst.position.y * scale, inst.position.z * scale
        inst.scale.x, inst.scale.y, inst.scale.z = inst.scale.x * scale, inst.scale.y * scale, inst.scale.z * scale
        host.set_instance(inst)
    cam = host.get_camera()
    cam.pos.x, cam.pos.y, cam.pos.z = cam.pos.x * scale, cam.pos.y * scale, cam.pos.z * scale
    host.set_camera(cam)
    return host


def test_oracle_aerial_perspective_adds_haze():
    plain = _with_sky_luts(_distant_scene(False).device_scene())
    hazy = _with_sky_luts(_distant_scene(True).device_scene())
    assert plain.sky_aerial_perspective == 0 and hazy.sky_aerial_perspective == 1
    a, _, _ = oracle_lib.render(plain, 0, 2)
    b, _, _ = oracle_lib.render(hazy, 0, 2)
    assert np.isfinite(b).all() and not np.array_equal(a, b)
    # the ground in the lower half of the frame is kilometres away: the air in front of it adds light, blue more than red
    lower_a, lower_b = a.reshape(3, H, W)[:, H // 2 + 4:], b.reshape(3, H, W)[:, H // 2 + 4:]
    gain = lower_b.mean(axis=(1, 2)) - lower_a.mean(axis=(1, 2))
    assert gain[2] > 0.0 and gain[2] > gain[0]
    # a constant-colour sky has no atmosphere: the switch does nothing there (device_manager.c:475)
    from luminary_amd import SKY_MODE_CONSTANT_COLOR
    c1 = oracle_lib.with_luts(_distant_scene(False, mode=SKY_MODE_CONSTANT_COLOR).device_scene())
    c2 = oracle_lib.with_luts(_distant_scene(True, mode=SKY_MODE_CONSTANT_COLOR).device_scene())
    assert np.array_equal(oracle_lib.render(c1, 0, 1)[0], oracle_lib.render(c2, 0, 1)[0])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [SKY_MODE_DEFAULT, SKY_MODE_HDRI])
def test_aerial_perspective_matches_the_oracle(mode):
    host = _distant_scene(True, mode=mode)
    plain = host.device_scene()
    view = oracle_lib.with_sky_hdri(plain) if mode == SKY_MODE_HDRI else _with_sky_luts(plain)
    core = Core(0)
    try:
        core.upload(view)
        core.set_pixels(None)
        core.reset_counters()
        core.render(0, 3, samples_per_pass=2)
        fm, sm = core.accumulators()
        ofm, osm, ocnt = oracle_lib.render(view, 0, 3)
        assert np.array_equal(fm, ofm), "first moment: %d of %d differ, max %g" % ((fm != ofm).sum(), fm.size, np.abs(fm - ofm).max())
        assert np.array_equal(sm, osm)
        assert core.query_counters()[:4] == [int(x) for x in ocnt[:4]]
    finally:
        core.close()
