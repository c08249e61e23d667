"""GPU parity tests: the HIP path (through the C ABI) against the oracle on the same seeded inputs.
Integer paths and, by the numerics contract of DESIGN.md, the float radiance are compared bit for bit."""
import numpy as np
import pytest

import oracle_lib
from luminary_amd import scenes
from luminary_amd.core import Core

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def core():
    c = Core(0)
    yield c
    c.close()


def _cornell(tmp_path_factory, w, h, bounces):
    d = tmp_path_factory.mktemp("cornell")
    return scenes.cornell_host(str(d), w, h, bounces)


def _assert_same(a, b, what):
    if not np.array_equal(a, b):
        diff = np.abs(a.astype(np.float64) - b.astype(np.float64))
        bad = int((a != b).sum())
        raise AssertionError("%s: %d of %d values differ, max abs diff %g" % (what, bad, a.size, diff.max()))


def test_lut_generation_matches_oracle(core, tmp_path_factory):
    """bsdf_lut.cuh:20-211 on the GPU == oracle tables (tests/golden/bsdf_luts.npz), all 67584 texels."""
    host = _cornell(tmp_path_factory, 16, 16, 1)
    core.upload(host.device_scene())  # LUT pointers are NULL -> generated on the GPU
    got = core.download_luts()
    want = oracle_lib.golden_luts()
    for k in ("conductor", "glossy", "dielectric", "dielectric_inv"):
        _assert_same(got[k], want[k], "lut " + k)


def _random_rays(n, seed, lo, hi):
    rng = np.random.RandomState(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


def test_trace_parity_cornell(core, tmp_path_factory):
    host = _cornell(tmp_path_factory, 16, 16, 1)
    view = oracle_lib.with_luts(host.device_scene())
    core.upload(view)
    o, d = _random_rays(20000, 1, -1.0, 2.0)
    got = core.trace_closest_host(o, d)
    want = oracle_lib.trace_closest(view, o, d, use_bvh=False)
    _assert_same(got, want, "closest hits (cornell, brute-force oracle)")


def test_trace_parity_instanced_scene(core):
    host = scenes.example_scene(64, 36, 2, sphere_segments=8, ground_res=16, num_objects=24, num_lights=4)
    view = oracle_lib.with_luts(host.device_scene())
    core.upload(view)
    o, d = _random_rays(50000, 2, -30.0, 30.0)
    o[:, 1] = np.abs(o[:, 1]) * 0.5 + 0.5
    ign = np.full((o.shape[0], 2), 0xFFFFFFFF, dtype=np.uint32)
    got = core.trace_closest_host(o, d, ign)
    want = oracle_lib.trace_closest(view, o, d, ign, use_bvh=True)
    _assert_same(got, want, "closest hits (instanced scene)")
    # second round from the hit points with the ignore handle set (what bounce rays do)
    hit = got[:, 0] != 0xFFFFFFFE
    t = got[:, 2].copy().view(np.float32)
    o2 = (o + d * np.where(hit, t, 0.0)[:, None]).astype(np.float32)
    ign2 = got[:, :2].copy()
    ign2[~hit] = 0xFFFFFFFF
    d2 = _random_rays(o.shape[0], 3, 0, 1)[1]
    got2 = core.trace_closest_host(o2, d2, ign2)
    want2 = oracle_lib.trace_closest(view, o2, d2, ign2, use_bvh=True)
    _assert_same(got2, want2, "closest hits with ignore handles")


@pytest.mark.parametrize("bounces,spp", [(1, 1), (4, 3)])
def test_render_parity_cornell(core, tmp_path_factory, bounces, spp):
    """BASELINE config 0 (Cornell .obj, 1 spp, 1 bounce) and a deeper variant: per-pixel moments identical to the oracle."""
    host = _cornell(tmp_path_factory, 64, 64, bounces)
    view = oracle_lib.with_luts(host.device_scene())
    core.upload(view)
    core.set_pixels(None)
    core.reset_counters()
    core.render(0, spp, samples_per_pass=1)
    fm, sm = core.accumulators()
    ofm, osm, ocnt = oracle_lib.render(view, 0, spp)
    _assert_same(fm, ofm, "first moment")
    _assert_same(sm, osm, "second moment")
    cnt = core.query_counters()  # visibility QUERIES: rays traced + ambient samples answered by the next closest hit (the oracle traces every one)
    assert cnt[0] == ocnt[0] and cnt[1] == ocnt[1] and cnt[2] == ocnt[2] and cnt[3] == ocnt[3], (cnt, ocnt)
    assert fm.max() > 0.0


def test_render_parity_instanced_scene(core):
    host = scenes.example_scene(96, 54, 8, sphere_segments=8, ground_res=16, num_objects=24, num_lights=6)
    view = oracle_lib.with_luts(host.device_scene())
    core.upload(view)
    core.set_pixels(None)
    core.render(3, 2, samples_per_pass=2)
    fm, sm = core.accumulators()
    ofm, osm, _ = oracle_lib.render(view, 3, 2)
    _assert_same(fm, ofm, "first moment (instanced scene, 8 bounces)")
    _assert_same(sm, osm, "second moment")


def test_batching_and_tiling_do_not_change_the_image(core, tmp_path_factory):
    """Size-independent property: any split of samples into passes and of pixels into tiles reproduces the same sums."""
    host = _cornell(tmp_path_factory, 80, 48, 3)
    view = oracle_lib.with_luts(host.device_scene())
    core.upload(view)
    core.set_pixels(None)
    core.render(0, 6, samples_per_pass=1)
    ref, ref_sm = core.accumulators()
    core.clear()
    core.render(0, 4, samples_per_pass=4)
    core.render(4, 2, samples_per_pass=2)
    a, a_sm = core.accumulators()
    _assert_same(a, ref, "samples-per-pass split")
    _assert_same(a_sm, ref_sm, "samples-per-pass split (second moment)")
    import bench
    full = np.zeros_like(ref)
    for rank in range(3):
        px = bench.tile_pixels(view.width, view.height, rank, 3, tile=16)
        core.set_pixels(px)
        core.render(0, 6, samples_per_pass=3)
        part, _ = core.accumulators()
        full[:, px] = part
    _assert_same(full, ref, "image-tile partition over 3 ranks")


def test_product_fails_loudly_without_scene(core):
    from luminary_amd.core import CoreError
    c = Core(0)
    with pytest.raises(CoreError):
        c.set_pixels(None)
    c.close()


@pytest.mark.parametrize("sky_mode,aperture,blades", [(2, 0.0, 0), (0, 0.05, 0), (2, 0.08, 6)])
def test_render_parity_material_zoo(core, sky_mode, aperture, blades):
    """Every material branch (translucent, coloured/plain transparency, partial opacity, metals), one-sided emitters, rotated and
    non-uniformly scaled instances, a light tree with node descent (320 light triangles), both sky modes and the lens aperture:
    moments and ray counters identical to the oracle."""
    host = scenes.zoo_scene(96, 64, 8, sky_mode=sky_mode, aperture=aperture, blades=blades)
    # sky mode DEFAULT is the procedural atmosphere with sun sampling: the oracle needs its two tables (the GPU is handed the same ones)
    view = oracle_lib.with_sky_luts(host.device_scene()) if sky_mode == 0 else oracle_lib.with_luts(host.device_scene())
    assert view.num_light_tree_nodes > 0, "the scene must force light-tree descent"
    core.upload(view)
    core.set_pixels(None)
    core.reset_counters()
    core.render(5, 3, samples_per_pass=3)
    fm, sm = core.accumulators()
    ofm, osm, ocnt = oracle_lib.render(view, 5, 3)
    _assert_same(fm, ofm, "first moment (material zoo)")
    _assert_same(sm, osm, "second moment (material zoo)")
    cnt = core.query_counters()
    assert cnt[:4] == [int(x) for x in ocnt[:4]], (cnt, ocnt)
    assert np.isfinite(fm).all() and fm.max() > 0.0


def test_trace_parity_material_zoo(core):
    host = scenes.zoo_scene(32, 32, 1)
    view = oracle_lib.with_luts(host.device_scene())
    core.upload(view)
    o, d = _random_rays(40000, 11, -12.0, 12.0)
    o[:, 1] = np.abs(o[:, 1]) * 0.4 + 0.2
    ign = np.full((o.shape[0], 2), 0xFFFFFFFF, dtype=np.uint32)
    got = core.trace_closest_host(o, d, ign)
    want = oracle_lib.trace_closest(view, o, d, ign, use_bvh=False)
    _assert_same(got, want, "closest hits (rotated, non-uniformly scaled instances) vs brute force")


def _full_size_frame_properties(core, host, min_triangles, width=1920, height=1080):
    """Size-independent properties of a 1920x1080, 8-bounce frame instead of a full oracle frame: a strided sample of pixels equals the
    oracle exactly (moments and, for those pixels alone, the ray counters), two sample ids rendered in one pass equal two passes, and
    a 3-way tile partition reproduces the full frame bit for bit (exact equality and a checksum of the bit patterns)."""
    view = oracle_lib.with_luts(host.device_scene())
    assert core_total_triangles(view) >= min_triangles
    core.upload(view)
    core.set_pixels(None)
    core.reset_counters()
    core.render(0, 2, samples_per_pass=2)
    full, full_sm = core.accumulators()
    cnt = core.counters()
    assert np.isfinite(full).all() and full.max() > 0.0 and cnt[0] > 2 * width * height
    checksum = int(full.view(np.uint32).astype(np.uint64).sum() + full_sm.view(np.uint32).astype(np.uint64).sum())

    px = np.arange(0, width * height, 977 * (width * height // (1920 * 1080)), dtype=np.uint32)  # 2123 pixels across the frame
    ofm, osm, ocnt = oracle_lib.render(view, 0, 2, pixels=px)
    _assert_same(full[:, px], ofm, "strided pixels of the full-size frame vs oracle")
    _assert_same(full_sm[px], osm, "second moment of the strided pixels")
    core.set_pixels(px)
    core.reset_counters()
    core.render(0, 2, samples_per_pass=2)
    sub, sub_sm = core.accumulators()
    _assert_same(sub, ofm, "strided pixels rendered alone vs oracle")
    assert core.query_counters()[:4] == [int(x) for x in ocnt[:4]], "ray counters of the strided pixels"

    core.set_pixels(None)
    core.render(0, 1, samples_per_pass=1)
    core.render(1, 1, samples_per_pass=1)
    two, two_sm = core.accumulators()
    _assert_same(two, full, "one sample id per pass vs two per pass")
    _assert_same(two_sm, full_sm, "second moment, one sample id per pass vs two per pass")

    import bench
    acc = np.zeros_like(full)
    acc_sm = np.zeros_like(full_sm)
    for rank in range(3):
        tiles = bench.tile_pixels(width, height, rank, 3)
        core.set_pixels(tiles)
        core.render(0, 2, samples_per_pass=2)
        part, part_sm = core.accumulators()
        acc[:, tiles] = part
        acc_sm[tiles] = part_sm
    assert int(acc.view(np.uint32).astype(np.uint64).sum() + acc_sm.view(np.uint32).astype(np.uint64).sum()) == checksum
    _assert_same(acc, full, "3-rank tile partition at full size")


def core_total_triangles(view):
    import ctypes
    off = np.ctypeslib.as_array(ctypes.cast(view.mesh_tri_offset, ctypes.POINTER(ctypes.c_uint32)), (view.num_meshes + 1,))
    return int(off[view.num_meshes])


def test_full_size_frame_properties(core):
    """BASELINE config 2 at its full size (Example-class scene, 1920x1080, 8 bounces, ~100 k triangles, 72 instances)."""
    _full_size_frame_properties(core, scenes.example_scene(1920, 1080, 8), 14_000)  # unique triangles; ~100 k once instanced


def test_full_size_frame_properties_4k(core):
    """BASELINE config 4's frame (Example-class scene at 3840x2160) on one GPU: 16.6 M paths per 2-sample pass, the same size-independent
    properties; the 3-way tile partition is what the 8-GPU run of that config does with 8."""
    _full_size_frame_properties(core, scenes.example_scene(3840, 2160, 8), 14_000, 3840, 2160)


def test_full_size_frame_properties_hall_1m(core):
    """BASELINE config 3, the scene the north-star target is quoted on: the 1 M-triangle Sponza-class hall at 1920x1080, 8 bounces."""
    _full_size_frame_properties(core, scenes.hall_scene(1920, 1080, 8), 1_000_000)


def test_full_size_frame_properties_scan_10m(core):
    """BASELINE config 5's scene at its stated size: the 10 M-triangle scanned-object class scene at 1920x1080, 8 bounces (single GPU
    here; the 8-GPU run partitions the same frame by tiles, which the 3-way partition of this test reproduces bit for bit)."""
    _full_size_frame_properties(core, scenes.scan_scene(1920, 1080, 8, triangles=10_000_000), 10_000_000)


@pytest.mark.parametrize("mode", [1, 2, 3, 4, 5])
def test_debug_shading_modes_match_the_oracle(core, mode):
    """settings.shading_mode ALBEDO / DEPTH / NORMAL / IDENTIFICATION / LIGHTS (geometry_process_tasks_debug, cuda/geometry.cuh:182-246;
    sky_process_tasks_debug, cuda/sky.cuh:635-665): one closest-hit pass and a colour per path, no random numbers beyond the pixel jitter.
    Checked on the material zoo (rotated, scaled instances; emitters; every material branch) with an open sky so that misses occur."""
    host = scenes.zoo_scene(96, 64, 8)
    st = host.get_settings()
    st.shading_mode = mode
    host.set_settings(st)
    view = oracle_lib.with_luts(host.device_scene())
    assert view.shading_mode == mode
    core.upload(view)
    core.set_pixels(None)
    core.reset_counters()
    core.render(0, 3, samples_per_pass=3)
    fm, sm = core.accumulators()
    ofm, osm, ocnt = oracle_lib.render(view, 0, 3)
    _assert_same(fm, ofm, "first moment, shading mode %d" % mode)
    _assert_same(sm, osm, "second moment, shading mode %d" % mode)
    cnt = core.counters()
    assert cnt[0] == int(ocnt[0]) == 3 * 96 * 64 and cnt[1] == 0 and cnt[2] == 0, "one closest-hit ray per path and nothing else"
    assert float(fm.max()) > 0.0


def test_ray_sorting_does_not_change_results(core):
    """Ray ordering (lumc_set_ray_sorting): tracing the closest-hit rays of depth >= 1 (mode 1) and the visibility rays (mode 2) in the order
    of a sort by origin cell and direction octant, or physically reordering the path queue by it at every depth (mode 3), gives bit-identical
    moments and counters - every path owns its slots."""
    host = scenes.example_scene(256, 144, 8, sphere_segments=10, ground_res=24, num_objects=32, num_lights=8)
    view = oracle_lib.with_luts(host.device_scene())
    core.upload(view)
    core.set_pixels(None)
    ref = None
    core.set_ambient_reuse(0)  # the reuse is off whenever rays are sorted: compare like with like (node visits and visibility rays are among the counters)
    try:
        for mode in (0, 1, 2, 3):
            core.set_ray_sorting(mode)
            core.clear()
            core.reset_counters()
            core.render(0, 4, samples_per_pass=2)
            fm, sm = core.accumulators()
            cnt = core.counters()[:10]
            if ref is None:
                ref = (fm, sm, cnt)
                ofm, osm, _ = oracle_lib.render(view, 0, 4, pixels=np.arange(0, 256 * 144, 37, dtype=np.uint32))
                _assert_same(fm[:, ::37], ofm, "unsorted render vs oracle (strided pixels)")
            else:
                _assert_same(fm, ref[0], "first moment, sort mode %d" % mode)
                _assert_same(sm, ref[1], "second moment, sort mode %d" % mode)
                assert cnt == ref[2], "counters, sort mode %d" % mode
    finally:
        core.set_ray_sorting(0)
        core.set_ambient_reuse(-1)


def test_render_parity_emission_textures(core):
    """Emission textures: textured emitters seen directly, sampled as lights (colour from the texel at the sampled point, alpha from the
    albedo texture), weighed in the light tree by their integrated texture maximum."""
    host = scenes.emissive_texture_scene(72, 48, 4)
    view = oracle_lib.with_luts(host.device_scene())
    assert view.num_lights == 5 and view.num_textures == 3
    core.upload(view)
    core.set_pixels(None)
    core.reset_counters()
    core.render(0, 6, samples_per_pass=3)
    fm, sm = core.accumulators()
    ofm, osm, ocnt = oracle_lib.render(view, 0, 6)
    _assert_same(fm, ofm, "first moment (emission textures)")
    _assert_same(sm, osm, "second moment (emission textures)")
    assert core.query_counters()[:4] == [int(x) for x in ocnt[:4]]


def test_render_parity_textured_scene(core):
    """Textures: albedo with gamma, alpha cut-outs in closest-hit and visibility rays, texture-driven coloured transparency,
    roughness and normal maps, a dangling texture handle: moments and ray counters identical to the oracle."""
    host = scenes.textured_scene(96, 64, 6)
    view = oracle_lib.with_luts(host.device_scene())
    assert view.num_textures == 5
    core.upload(view)
    core.set_pixels(None)
    core.reset_counters()
    core.render(2, 3, samples_per_pass=3)
    fm, sm = core.accumulators()
    ofm, osm, ocnt = oracle_lib.render(view, 2, 3)
    _assert_same(fm, ofm, "first moment (textured scene)")
    _assert_same(sm, osm, "second moment (textured scene)")
    assert core.query_counters()[:4] == [int(x) for x in ocnt[:4]]
    # closest hits through the cut-outs agree with brute force as well
    o, d = _random_rays(30000, 21, -9.0, 9.0)
    o[:, 1] = np.abs(o[:, 1]) * 0.3 + 0.1
    ign = np.full((o.shape[0], 2), 0xFFFFFFFF, dtype=np.uint32)
    _assert_same(core.trace_closest_host(o, d, ign), oracle_lib.trace_closest(view, o, d, ign, use_bvh=False), "closest hits with alpha cut-outs")


@pytest.mark.parametrize("kind", ["empty", "no_lights", "degenerate", "one_triangle"])
def test_render_parity_edge_scenes(core, kind):
    """Borders of the input domain: no geometry, no lights, zero-area / duplicated triangles with a collapsed instance, a single
    triangle. Moments and ray counters identical to the oracle."""
    host = scenes.edge_scene(kind, 48, 32, 4)
    view = oracle_lib.with_luts(host.device_scene())
    core.upload(view)
    core.set_pixels(None)
    core.reset_counters()
    core.render(0, 3, samples_per_pass=2)  # 2 + 1: a ragged last pass
    fm, sm = core.accumulators()
    ofm, osm, ocnt = oracle_lib.render(view, 0, 3)
    _assert_same(fm, ofm, "first moment (%s)" % kind)
    _assert_same(sm, osm, "second moment (%s)" % kind)
    assert core.query_counters()[:4] == [int(x) for x in ocnt[:4]]
    if kind == "empty":
        assert np.allclose(fm / 3.0, np.array([[0.4], [0.5], [0.7]], dtype=np.float32), rtol=1e-6)  # the constant sky, nothing else


@pytest.mark.parametrize("bounces", [0, 63])
def test_render_parity_depth_limits(core, tmp_path_factory, bounces):
    """max_ray_depth 0 (camera rays and their direct light only) and 63 (the largest value the 6-bit depth field holds)."""
    host = _cornell(tmp_path_factory, 40, 30, bounces)
    view = oracle_lib.with_luts(host.device_scene())
    core.upload(view)
    core.set_pixels(None)
    core.render(1, 2, samples_per_pass=2)
    fm, sm = core.accumulators()
    ofm, osm, _ = oracle_lib.render(view, 1, 2)
    _assert_same(fm, ofm, "first moment (max depth %d)" % bounces)
    _assert_same(sm, osm, "second moment (max depth %d)" % bounces)


def test_empty_pixel_set_and_single_pixel(core, tmp_path_factory):
    host = _cornell(tmp_path_factory, 40, 30, 2)
    view = oracle_lib.with_luts(host.device_scene())
    core.upload(view)
    one = np.array([40 * 15 + 20], dtype=np.uint32)
    core.set_pixels(one)
    core.render(0, 5, samples_per_pass=3)
    fm, sm = core.accumulators()
    ofm, osm, _ = oracle_lib.render(view, 0, 5, pixels=one)
    _assert_same(fm, ofm, "single pixel")
    _assert_same(sm, osm, "single pixel (second moment)")
