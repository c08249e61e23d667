"""Oracle self-checks on the CPU: BVH == brute force, determinism, partition invariance, LUT fixture, energy sanity."""
import ctypes as C

import numpy as np

import oracle_lib
from luminary_amd import scenes
from luminary_amd.distributed import tile_pixels


def _cornell(tmp_path, w=40, h=30, bounces=3):
    return oracle_lib.with_luts(scenes.cornell_host(str(tmp_path), w, h, bounces).device_scene())


def test_bvh_equals_brute_force(tmp_path):
    v = _cornell(tmp_path)
    a = oracle_lib.render(v, 0, 3, use_bvh=True)
    b = oracle_lib.render(v, 0, 3, use_bvh=False)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert a[0].max() > 0 and np.isfinite(a[0]).all()


def test_tile_partition_reproduces_the_frame(tmp_path):
    v = _cornell(tmp_path)
    full, full_sm, _ = oracle_lib.render(v, 2, 2)
    out = np.zeros_like(full)
    seen = np.zeros(full.shape[1], dtype=np.int32)
    for rank in range(4):
        px = tile_pixels(v.width, v.height, rank, 4, tile=8)
        part, _, _ = oracle_lib.render(v, 2, 2, pixels=px)
        out[:, px] = part
        seen[px] += 1
    assert (seen == 1).all()
    assert np.array_equal(out, full)


def test_samples_are_additive(tmp_path):
    v = _cornell(tmp_path)
    a, _, _ = oracle_lib.render(v, 0, 4)
    b, _, _ = oracle_lib.render(v, 0, 2)
    c, _, _ = oracle_lib.render(v, 2, 2)
    # float sums in sample order: (s0+s1)+(s2+s3) differs from ((s0+s1)+s2)+s3 only by rounding
    assert np.allclose(a, b + c, rtol=1e-5, atol=1e-6)


def test_lut_fixture_matches_the_oracle_on_sampled_texels():
    luts = oracle_lib.golden_luts()
    bn = oracle_lib.bluenoise()
    L = oracle_lib.lib()
    for table, name, texels in [(0, "conductor", [33, 500, 1023]), (1, "glossy", [40, 700]), (2, "dielectric", [5000]), (3, "dielectric_inv", [20000])]:
        for t in texels:
            out = np.zeros(1, np.uint16)
            L.oracle_generate_lut(bn.ctypes.data_as(C.c_void_p), C.c_int(table), C.c_uint32(t), C.c_uint32(1), luts["conductor"].ctypes.data_as(C.c_void_p),
                                  out.ctypes.data_as(C.c_void_p))
            assert out[0] == luts[name][t], (name, t)
    # directional albedo of a rough conductor is below one and decreases with roughness at normal incidence
    c = luts["conductor"].reshape(32, 32)
    assert c[31, 31] < c[8, 31] <= 65535


def test_white_furnace_is_bounded(tmp_path):
    """Closed white diffuse box lit only by its emitter: radiance stays finite and the mean is positive."""
    v = _cornell(tmp_path, 24, 24, 6)
    fm, _, cnt = oracle_lib.render(v, 0, 4)
    assert np.isfinite(fm).all() and fm.mean() > 0.01
    assert cnt[0] >= 24 * 24 * 4 and cnt[1] > 0 and cnt[3] > 0


def test_material_zoo_bvh_equals_brute_force_and_descends_the_light_tree():
    """The zoo scene covers translucent/transparent/metallic materials, one-sided emitters, transformed instances and a light tree
    with inner nodes; the oracle's two intersectors must agree on it and every ray class must occur."""
    v = oracle_lib.with_luts(scenes.zoo_scene(48, 32, 6).device_scene())
    assert v.num_lights > 128 and v.num_light_tree_nodes > 0
    a = oracle_lib.render(v, 0, 2, use_bvh=True)
    b = oracle_lib.render(v, 0, 2, use_bvh=False)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert np.isfinite(a[0]).all() and a[0].max() > 0
    assert a[2][0] > 0 and a[2][1] > 0 and a[2][2] > 0  # closest, shadow and light-BVH rays all present


def test_textures_change_the_image_and_cut_outs_let_rays_through():
    """Oracle self-check for the texture path: BVH == brute force on the textured scene, the alpha-0 texels of the fence are holes for
    closest-hit rays, and removing the textures changes the image."""
    host = scenes.textured_scene(48, 32, 4)
    v = oracle_lib.with_luts(host.device_scene())
    a = oracle_lib.render(v, 0, 2, use_bvh=True)
    b = oracle_lib.render(v, 0, 2, use_bvh=False)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.isfinite(a[0]).all()
    # rays straight at the fence (z = 2 plane, x in [-4, 4], y in [0, 3]) from the camera side: some pass through the holes
    n = 4000
    rng = np.random.RandomState(0)
    o = np.stack([rng.uniform(-3.9, 3.9, n), rng.uniform(0.1, 2.9, n), np.full(n, 6.0)], axis=1).astype(np.float32)
    d = np.tile(np.array([0.0, 0.0, -1.0], dtype=np.float32), (n, 1))
    hits = oracle_lib.trace_closest(v, o, d, np.full((n, 2), 0xFFFFFFFF, dtype=np.uint32), use_bvh=True)
    t = hits[:, 2].copy().view(np.float32)
    on_fence = np.isclose(t, 4.0, atol=1e-4)
    assert 0.5 < on_fence.mean() < 0.98  # slats and rails are hit; only texels whose filtered alpha is exactly 0 are holes
    untextured = oracle_lib.with_luts(host.device_scene())
    untextured.num_textures = 0
    c = oracle_lib.render(untextured, 0, 2)
    assert not np.array_equal(a[0], c[0])


def test_emission_textures_light_the_scene_and_shape_the_light_tree():
    """Emission textures (map_Ke): a triangle whose part of the texture is black is no light, a dangling texture handle emits nothing,
    the screens are visible (surface context) and light the floor (light sampling)."""
    host = scenes.emissive_texture_scene(48, 32, 3)
    v = oracle_lib.with_luts(host.device_scene())
    # lit half of the first screen (1 triangle), both triangles of the second screen, the constant emitter (2): the dark half and the
    # dangling screen are left out of the light tree (device_light.c:2082)
    assert v.num_lights == 5
    a = oracle_lib.render(v, 0, 4, use_bvh=True)
    b = oracle_lib.render(v, 0, 4, use_bvh=False)
    assert np.array_equal(a[0], b[0]) and np.isfinite(a[0]).all()
    img = a[0].reshape(3, 32, 48)
    assert img[:, 24:, :].mean() > 0.0, "the floor receives light although the sky is black"
    dark = oracle_lib.with_luts(host.device_scene())
    dark.num_textures = 0   # without the textures the screens are black: only the small constant emitter is left
    c = oracle_lib.render(dark, 0, 4)
    assert c[0].sum() < 0.5 * a[0].sum()
