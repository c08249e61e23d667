"""Ambient-visibility reuse (include/lum_core.h lumc_set_ambient_reuse; kernels.h above TraceQuery; the depth loop in core.hip).

Under a constant-colour or panorama sky the ambient sample of a vertex runs along its bounce direction (cuda/direct_lighting.cuh:388-405), i.e. along the
closest-hit ray the path traces at the next depth anyway. Where that ray's nearest hit is opaque, or it leaves the scene, the visibility is known and no
visibility ray is traced; a transparent or textured nearest hit, a skipped alpha cut-out, and paths that end at the vertex still trace theirs.
The reference's ambient ray starts at the hit point and uses the direction after its 2 x 32-bit packing, a last bit away from the ray the path goes on along.
The EXACT flavour therefore takes "blocked" only after testing the ambient ray itself against the triangle the closest hit found and traces every sample
whose ray found nothing: its images must be IDENTICAL with the reuse on and off, and to the oracle (it is off by default there: the proof's gathers cost
more than the cheap rays they save). The FAST
flavour takes the closest hit's word: its images may differ from the traced ones in a handful of edge pixels. In both, the ray counters add up exactly."""
import numpy as np
import pytest

import oracle_lib
from luminary_amd import scenes
from luminary_amd.core import CNT_AMBIENT_DEFERRED, CNT_AMBIENT_FALLBACK, CNT_LIGHT_BVH, CNT_SHADOW, CNT_TRACE, CNT_VERTICES, Core


def test_the_c_abi_exports_the_switch():
    from luminary_amd import _lib
    lib = _lib()
    assert hasattr(lib, "lumc_set_ambient_reuse") and hasattr(lib, "lumc_get_ambient_reuse")


def _render(core, mode, spp=8, batch=4):
    core.set_ambient_reuse(mode)
    core.set_pixels(None)
    core.reset_counters()
    core.render(0, spp, samples_per_pass=batch)
    fm, sm = core.accumulators()
    return fm.copy(), sm.copy(), core.counters()


def _scene(name, tmp):
    if name == "cornell":        # the generated box has a black sky (no ambient sample at all): give it one, the box is open at the front
        host = scenes.cornell_host(str(tmp), 96, 96, 6)
        sky = host.get_sky()
        sky.constant_color.r, sky.constant_color.g, sky.constant_color.b = 0.5, 0.6, 0.8
        host.set_sky(sky)
        return host
    if name == "zoo":            # glass, coloured transparency, metals, hundreds of emitters: transparent first hits -> the fallback pass
        return scenes.zoo_scene(96, 64, 8)
    if name == "textured":       # alpha cut-outs and texture-driven transparency: the cut-out flag -> the fallback pass
        return scenes.textured_scene(96, 64, 6)
    if name == "example":
        return scenes.example_scene(160, 96, 6, sphere_segments=8, ground_res=12, num_objects=16, num_lights=4)
    if name == "no_lights":
        return scenes.edge_scene("no_lights", 64, 48, 4)
    raise ValueError(name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "zoo", "textured", "example", "no_lights"])
def test_reuse_in_the_exact_flavour_is_bit_identical_to_tracing(name, tmp_path):
    host = _scene(name, tmp_path)
    view = oracle_lib.with_luts(host.device_scene())
    core = Core(0)
    try:
        core.set_flavour("exact")
        core.upload(view)
        core.set_ambient_reuse(-1)
        assert not core.ambient_reuse, "off by default in the exact flavour (the proof costs more than the rays it saves)"
        fm0, sm0, c0 = _render(core, 0)
        fm1, sm1, c1 = _render(core, 1)
        fm2, _, _ = _render(core, 0)
    finally:
        core.close()
    assert np.array_equal(fm0, fm2), "switching back and forth must not change the traced result"
    assert c0[CNT_AMBIENT_DEFERRED] == 0 and c0[CNT_AMBIENT_FALLBACK] == 0
    assert np.array_equal(fm1, fm0) and np.array_equal(sm1, sm0), "the exact flavour's reuse only takes what it can prove: identical images"
    # the paths are the same paths: closest-hit rays, light queries and vertices do not change; every ambient query is either traced or answered
    for k in (CNT_TRACE, CNT_LIGHT_BVH, CNT_VERTICES):
        assert c1[k] == c0[k]
    assert c1[CNT_AMBIENT_DEFERRED] > 0, "nothing was deferred: the reuse did not run"
    assert c1[CNT_SHADOW] + c1[CNT_AMBIENT_DEFERRED] - c1[CNT_AMBIENT_FALLBACK] == c0[CNT_SHADOW]
    assert c1[CNT_AMBIENT_FALLBACK] > 0, "samples whose ray found nothing are traced (a miss cannot be proved from another ray)"
    if name in ("cornell", "example"):
        assert c1[CNT_AMBIENT_FALLBACK] < c1[CNT_AMBIENT_DEFERRED], "opaque hits beyond eps are answered without a trace"
    ofm, osm, ocnt = oracle_lib.render(view, 0, 8)
    assert np.array_equal(fm1, ofm) and np.array_equal(sm1, osm), "... and identical to the oracle"
    assert c1[CNT_SHADOW] + c1[CNT_AMBIENT_DEFERRED] - c1[CNT_AMBIENT_FALLBACK] == int(ocnt[1])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "zoo", "textured", "example", "no_lights"])
def test_reuse_in_the_fast_flavour_equals_tracing_up_to_edge_pixels(name, tmp_path):
    host = _scene(name, tmp_path)
    view = oracle_lib.with_luts(host.device_scene())
    core = Core(0)
    try:
        core.set_flavour("fast")
        core.upload(view)
        assert core.ambient_reuse
        fm0, sm0, c0 = _render(core, 0)
        fm1, sm1, c1 = _render(core, 1)
    finally:
        core.close()
    for k in (CNT_TRACE, CNT_LIGHT_BVH, CNT_VERTICES):
        assert c1[k] == c0[k]
    assert c1[CNT_AMBIENT_DEFERRED] > 0
    assert c1[CNT_SHADOW] + c1[CNT_AMBIENT_DEFERRED] - c1[CNT_AMBIENT_FALLBACK] == c0[CNT_SHADOW]
    if name in ("zoo", "textured"):
        assert c1[CNT_AMBIENT_FALLBACK] > 0, "this scene has transparent / cut-out first hits: the fallback pass must have traced some"
    if name in ("cornell", "no_lights"):
        assert c1[CNT_AMBIENT_FALLBACK] <= c1[CNT_AMBIENT_DEFERRED] // 1000, "opaque scene: (almost) every deferred sample is answered by the closest hit"
    # the images: identical except where the last bit of the direction decides whether a ray grazes an edge (and, with the fused resolve - the default -,
    # up to the rounding of the order in which an undecided sample's vertex and the next depth's emission are added)
    differing = int((np.abs(fm1 - fm0) > 4e-6 * np.maximum(np.abs(fm0), np.abs(fm1)) + 1e-7).any(axis=0).sum())
    assert differing <= max(4, fm0.shape[1] // 500), "%d of %d pixels differ" % (differing, fm0.shape[1])
    assert abs(float(fm1.sum()) - float(fm0.sum())) <= 1e-4 * float(fm0.sum())
    assert np.isfinite(fm1).all() and np.isfinite(sm1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "zoo", "textured", "example", "no_lights", "hdri"])
def test_the_fused_resolve_gives_the_separate_kernels_sums(name, tmp_path):
    """lumc_set_fused_resolve (fast flavour with the reuse): the resolve of a depth done by the next depth's shading kernel - every entry resolves the vertex
    it continues - and by k_resolve_ended for the vertices whose path ended, against the separate k_resolve_reuse: the same loads, the same sums in the same
    order, so identical images - except for the samples the closest hit cannot decide (transparent or cut-out first hits), whose vertex is summed after the
    next depth's emission instead of before it: a last-bit difference in those pixels. Counters identical in every case."""
    if name == "hdri":           # the panorama sky: the sun is the fourth record of a vertex, its visibility word the fourth kind
        from luminary_amd import SKY_MODE_HDRI
        host = scenes.example_scene(128, 80, 5, sphere_segments=8, ground_res=12, num_objects=12, num_lights=3)
        sky = host.get_sky()
        sky.mode = SKY_MODE_HDRI
        host.set_sky(sky)
    else:
        host = _scene(name, tmp_path)
    view = oracle_lib.with_luts(host.device_scene())
    core = Core(0)
    try:
        core.set_flavour("fast")
        core.upload(view)
        assert core.ambient_reuse
        core.set_fused_resolve(False)
        fm0, sm0, c0 = _render(core, -1)
        core.set_fused_resolve(True)
        fm1, sm1, c1 = _render(core, -1)
        core.set_fused_resolve(2)   # the ended paths' vertices by k_resolve_ended instead of by the next depth's shading kernel: the same sums per result slot
        fm3, sm3, c3 = _render(core, -1)
        core.set_fused_resolve(False)
        fm2, _, _ = _render(core, -1)
    finally:
        core.close()
    assert np.array_equal(fm0, fm2)
    assert list(c3) == list(c1)
    # (k_resolve_ended forms a vertex's sum through resolve_records, the shading kernel through the written-out form: under the fast flavour's contraction a
    # product with a fractional visibility - transparent or cut-out occluders - may round once less in one of them)
    assert np.allclose(fm3, fm1, rtol=2e-6, atol=1e-7) and np.allclose(sm3, sm1, rtol=4e-6, atol=1e-7)
    if name in ("cornell", "example", "no_lights"):
        assert np.array_equal(fm3, fm1) and np.array_equal(sm3, sm1), "opaque scenes: every visibility is 0 or 1, the sums are the same bits"
    assert list(c1) == list(c0), "the same rays, vertices and fallbacks either way"
    assert c1[CNT_AMBIENT_DEFERRED] > 0
    if c1[CNT_AMBIENT_FALLBACK] == 0:
        assert np.array_equal(fm1, fm0) and np.array_equal(sm1, sm0)
    else:
        differing = int((np.abs(fm1 - fm0).max(axis=0) > 0).sum())
        assert differing <= c1[CNT_AMBIENT_FALLBACK], "only pixels with an undecided sample may differ (%d pixels, %d samples)" % (differing, c1[CNT_AMBIENT_FALLBACK])
        assert np.allclose(fm1, fm0, rtol=2e-6, atol=1e-7), "... and only in the rounding of the order of the sums"
    assert np.isfinite(fm1).all()


@pytest.mark.gpu
def test_the_fused_resolve_over_depths_pixel_subsets_and_growing_passes(tmp_path):
    """The fused resolve's buffers follow the work buffers (a larger pass reallocates both), its queue rotation any maximum depth (0: nothing to fuse; 1: one
    fused depth; 7), and a pixel subset renders what the full frame renders for those pixels - each time the separate kernels' image, bit for bit (an opaque
    scene: no undecided samples)."""
    def host_with_depth(depth):
        host = scenes.cornell_host(str(tmp_path / ("d%d" % depth)), 64, 48, depth)
        sky = host.get_sky()
        sky.constant_color.r, sky.constant_color.g, sky.constant_color.b = 0.5, 0.6, 0.8
        host.set_sky(sky)
        return host

    for depth in (0, 1, 2, 7):
        view = oracle_lib.with_luts(host_with_depth(depth).device_scene())
        core = Core(0)
        try:
            core.set_flavour("fast")
            core.upload(view)
            images = {}
            for fused in (False, True):
                core.set_fused_resolve(fused)
                frames = []
                for pixels, spp, batch in ((np.arange(5, 64 * 48, 7, dtype=np.uint32), 3, 1), (None, 4, 2), (None, 16, 16), (np.arange(0, 64 * 48, 2, dtype=np.uint32), 6, 3)):
                    core.set_pixels(pixels)   # the third render is the largest pass so far: the buffers grow between renders
                    core.reset_counters()
                    core.render(0, spp, samples_per_pass=batch)
                    frames.append((core.accumulators()[0].copy(), list(core.counters())))
                images[fused] = frames
        finally:
            core.close()
        for (fm0, c0), (fm1, c1) in zip(images[False], images[True]):
            assert c0 == c1, "depth %d" % depth
            assert c1[CNT_AMBIENT_FALLBACK] == 0
            assert np.array_equal(fm0, fm1), "depth %d" % depth
        full, sub = images[True][1][0], images[True][0][0]
        assert full.shape[1] == 64 * 48 and sub.shape[1] == len(range(5, 64 * 48, 7))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["dark", "bright"])
def test_the_ended_paths_vertices_in_the_next_depths_shading_kernel_at_extreme_ratios(kind, tmp_path):
    """k_shade of depth d + 1 takes depth d's ended vertices as further input, their rounds spread between its queue's rounds (kernels.h entry_at). Dark walls: almost
    every path ends at its first vertices, so the listed vertices outnumber the queue's entries (one listed round per queue round, and a tail of listed rounds); bright
    walls and no roulette to speak of: hardly any path ends early (many queue rounds per listed round, and depths without a single listed vertex). Against the
    separate kernels, bit for bit (opaque scene), for pass sizes that leave partial rounds."""
    host = scenes.cornell_host(str(tmp_path), 80, 56, 6)
    sky = host.get_sky()
    sky.constant_color.r, sky.constant_color.g, sky.constant_color.b = 0.5, 0.6, 0.8
    host.set_sky(sky)
    for i in range(8):
        try:
            m = host.get_material(i)
        except Exception:  # noqa: BLE001 - past the scene's last material
            break
        if m.emission_active:
            continue
        v = 0.02 if kind == "dark" else 0.98
        m.albedo.r, m.albedo.g, m.albedo.b = v, v, v
        host.set_material(i, m)
    if kind == "bright":
        cam = host.get_camera()
        cam.russian_roulette_threshold = 1e-6
        host.set_camera(cam)
    view = oracle_lib.with_luts(host.device_scene())
    core = Core(0)
    try:
        core.set_flavour("fast")
        core.upload(view)
        frames = {}
        for mode in (0, 1, 2):
            core.set_fused_resolve(mode)
            out = []
            for spp, batch in ((5, 5), (3, 1), (7, 4)):
                core.set_pixels(None)
                core.reset_counters()
                core.render(0, spp, samples_per_pass=batch)
                out.append((core.accumulators()[0].copy(), list(core.counters())))
            frames[mode] = out
    finally:
        core.close()
    for a, b, c in zip(frames[0], frames[1], frames[2]):
        assert a[1] == b[1] == c[1]
        assert b[1][CNT_AMBIENT_FALLBACK] == 0
        assert np.array_equal(a[0], b[0]) and np.array_equal(c[0], b[0])
        assert np.isfinite(b[0]).all() and float(b[0].sum()) > 0.0


@pytest.mark.gpu
def test_default_by_flavour_and_scene(tmp_path):
    """fast: on for plain scenes; off with fog / under the procedural sky / with ray sorting; exact: off unless asked for."""
    from luminary_amd import SKY_MODE_DEFAULT
    host = scenes.cornell_host(str(tmp_path), 32, 32, 2)
    core = Core(0)
    try:
        core.set_flavour("fast")
        core.upload(oracle_lib.with_luts(host.device_scene()))
        assert core.ambient_reuse
        core.set_ray_sorting(1)
        assert not core.ambient_reuse
        core.set_ray_sorting(0)
        core.set_flavour("exact")
        assert not core.ambient_reuse
        core.set_ambient_reuse(1)
        assert core.ambient_reuse
        core.set_ambient_reuse(-1)
        core.set_flavour("fast")
        fog = host.get_fog()
        fog.active, fog.density = True, 10.0
        host.set_fog(fog)
        core.upload(oracle_lib.with_luts(host.device_scene()))
        assert not core.ambient_reuse
        fog.active = False
        host.set_fog(fog)
        sky = host.get_sky()
        sky.mode = SKY_MODE_DEFAULT
        host.set_sky(sky)
        core.upload(oracle_lib.with_luts(host.device_scene()))
        assert not core.ambient_reuse
    finally:
        core.close()


@pytest.mark.gpu
def test_fast_flavour_with_and_without_reuse(tmp_path):
    """The benchmarked configuration: fast flavour, reuse on (its default) against reuse off - same paths, same counters but the visibility rays, images within
    the handful of edge pixels; under the panorama sky (HDRI mode) as well."""
    from luminary_amd import SKY_MODE_HDRI
    for hdri in (False, True):
        host = scenes.example_scene(160, 96, 6, sphere_segments=8, ground_res=12, num_objects=16, num_lights=4)
        if hdri:
            sky = host.get_sky()
            sky.mode = SKY_MODE_HDRI
            host.set_sky(sky)
        core = Core(0)
        try:
            core.set_flavour("fast")
            core.upload(oracle_lib.with_luts(host.device_scene()))
            fm0, _, c0 = _render(core, 0)
            fm1, _, c1 = _render(core, -1)
        finally:
            core.close()
        assert c1[CNT_AMBIENT_DEFERRED] > 0 and c0[CNT_AMBIENT_DEFERRED] == 0
        assert c1[CNT_TRACE] == c0[CNT_TRACE] and c1[CNT_VERTICES] == c0[CNT_VERTICES]
        assert c1[CNT_SHADOW] + c1[CNT_AMBIENT_DEFERRED] - c1[CNT_AMBIENT_FALLBACK] == c0[CNT_SHADOW]
        differing = int((np.abs(fm1 - fm0).max(axis=0) > 0).sum())
        assert differing <= max(4, fm0.shape[1] // 500), "%d of %d pixels differ (hdri=%s)" % (differing, fm0.shape[1], hdri)


@pytest.mark.gpu
@pytest.mark.parametrize("adaptive", [False, True])
def test_the_fused_resolve_behind_the_host_api_tiled_and_adaptive(tmp_path, monkeypatch, adaptive):
    """The same comparison one level up: the unchanged host API in the fast flavour (what a Luminary user gets), the frame tiled over three device slots, uniform
    and adaptive sampling - LUM_FUSED_RESOLVE=1 (default) against 0, identical images and ray counters (an opaque scene: no undecided samples)."""
    monkeypatch.setenv("LUM_FLAVOUR", "fast")
    monkeypatch.setenv("LUM_FAKE_DEVICES", "3")
    monkeypatch.setenv("LUM_MAX_DEVICES", "8")

    def render(tag, fused):
        monkeypatch.setenv("LUM_FUSED_RESOLVE", "1" if fused else "0")
        host = scenes.cornell_host(str(tmp_path / tag), 96, 80, 4)
        sky = host.get_sky()
        sky.constant_color.r, sky.constant_color.g, sky.constant_color.b = 0.5, 0.6, 0.8
        host.set_sky(sky)
        if adaptive:
            s = host.get_settings()
            s.enable_adaptive_sampling = True
            s.adaptive_sampling_max_sampling_rate, s.adaptive_sampling_avg_sampling_rate, s.adaptive_sampling_update_interval = 8, 2, 2
            host.set_settings(s)
        assert host.get_device_count() == 3
        host.set_output_properties(96, 80)
        host.render(6)
        fm, sm = host.accumulators()
        img, n, _ = host.get_image(host.acquire_output())
        out = (fm.copy(), sm.copy(), img.copy(), n, list(host.ray_counters()[:4]))
        host.close()
        return out

    a, b = render("separate", False), render("fused", True)
    assert a[3] == b[3] and a[4] == b[4]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [5, 7])
def test_the_fused_resolve_after_a_pass_that_reordered_the_queue(tmp_path, depth):
    """Ray-sorting mode 3 makes a path queue's planes change places with the reorder pass's at every depth; after ONE such pass a queue that was reordered an odd
    number of times lives in other planes than before. The fused resolve (off while rays are sorted) reads the previous depth's queue through records in device
    memory, which must follow: fused pass, sorted pass (other sample ids), the first fused pass again == the first. (Without the refresh: differences of 10+.)"""
    host = scenes.cornell_host(str(tmp_path), 96, 96, depth)
    sky = host.get_sky()
    sky.constant_color.r, sky.constant_color.g, sky.constant_color.b = 0.5, 0.6, 0.8
    host.set_sky(sky)
    core = Core(0)
    try:
        core.set_flavour("fast")
        core.upload(oracle_lib.with_luts(host.device_scene()))
        core.set_fused_resolve(True)

        def one_pass(first):
            core.set_pixels(None)
            core.render(first, 4, samples_per_pass=4)
            return core.accumulators()[0].copy()

        first = one_pass(0)
        core.set_ray_sorting(3)
        one_pass(100)
        core.set_ray_sorting(0)
        again = one_pass(0)
    finally:
        core.close()
    assert np.array_equal(first, again)
