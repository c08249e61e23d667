"""Particles (optix_kernel_raytrace.cu:97-131, cuda/particle.cuh, device_particle.c): quads in a unit cell tiled 25^3 times, seen by delta
paths. CPU: the host layer's generator and the oracle's lattice tracer against independent numpy; GPU: HIP == oracle bit for bit."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from luminary_amd import SKY_MODE_CONSTANT_COLOR, SKY_MODE_DEFAULT, SKY_MODE_HDRI, scenes

L = oracle_lib.lib()


def _with_particles(host, count=2048, size=25.0, scale=4.0, speed=0.0, albedo=(0.9, 0.8, 0.7), seed=3, diameter=50.0):
    p = host.get_particles()
    p.active, p.count, p.size, p.scale, p.speed, p.seed, p.phase_diameter = True, count, size, scale, speed, seed, diameter
    p.albedo.r, p.albedo.g, p.albedo.b = albedo
    host.set_particles(p)
    return host


def _view(host):
    sky = host.get_sky()
    if sky.mode == SKY_MODE_HDRI and sky.hdri_dim > 32:
        sky.hdri_dim, sky.hdri_samples = 32, 3
        host.set_sky(sky)
    plain = host.device_scene()
    if sky.mode == SKY_MODE_HDRI:
        return oracle_lib.with_sky_hdri(plain)
    if sky.mode == SKY_MODE_DEFAULT:
        return oracle_lib.with_sky_luts(oracle_lib.with_luts(plain))
    return oracle_lib.with_luts(plain)


def _squares16(offset):
    """random_uint16_t (random.cuh:196-211, :297-299) in numpy."""
    key = np.uint64(0xfcbd6e15)
    m = np.uint64(0xFFFFFFFF)
    c = offset.astype(np.uint64)
    x = (c * key) & m
    y = x.copy()
    z = (y + key) & m
    x = (x * x + y) & m; x = ((x >> np.uint64(16)) | (x << np.uint64(16))) & m
    x = (x * x + z) & m; x = ((x >> np.uint64(16)) | (x << np.uint64(16))) & m
    return (((x * x + y) & m) >> np.uint64(16)).astype(np.uint32)


def test_generated_particles_follow_the_generator(tmp_path):
    """particle_generate (cuda/particle.cuh:165-211) restated with numpy: centres from the 16-bit Squares generator, a square of half-size
    0.001 * size * (1 + variation * r) in the plane orthogonal to a uniformly drawn normal, two triangles sharing the a01-a10 diagonal."""
    host = _with_particles(scenes.cornell_host(str(tmp_path), 8, 8, 1), count=500, size=20.0, seed=11)
    view = host.device_scene()
    assert view.particles_active == 1 and view.particles_count == 500
    verts = np.ctypeslib.as_array(C.cast(view.particle_vertices, C.POINTER(C.c_float)), shape=(500, 6, 4)).copy()
    normals = np.ctypeslib.as_array(C.cast(view.particle_normals, C.POINTER(C.c_float)), shape=(500, 4)).copy()
    ids = np.arange(500, dtype=np.uint32)
    noise = lambda k: (_squares16(np.uint32(11) + ids * np.uint32(6) + np.uint32(k)).astype(np.float64) * 2 ** -16)  # 0x3F800000 | v << 7, minus 1
    centre = np.stack([noise(0), noise(1), noise(2)], axis=1)
    quad_centre = verts[:, :4, :3].astype(np.float64).mean(axis=1)
    assert np.allclose(quad_centre, centre, atol=1e-6)
    r1, r2 = 2 * noise(3) - 1, noise(4)
    n = np.stack([np.sqrt(1 - r1 ** 2) * np.cos(2 * np.pi * r2), np.sqrt(1 - r1 ** 2) * np.sin(2 * np.pi * r2), r1], axis=1)
    assert np.allclose(np.abs((normals[:, :3] * n).sum(axis=1)), 1.0, atol=1e-4), "the quad's normal is the drawn direction up to sign"
    half = 0.001 * 20.0 * (1 + 0.1 * (2 * noise(5) - 1))
    e1 = verts[:, 1, :3] - verts[:, 0, :3]; e2 = verts[:, 2, :3] - verts[:, 0, :3]
    assert np.allclose(np.linalg.norm(e1, axis=1), 2 * half, rtol=1e-3) and np.allclose(np.linalg.norm(e2, axis=1), 2 * half, rtol=1e-3)
    assert np.allclose((e1 * e2).sum(axis=1), 0.0, atol=1e-7)
    assert np.array_equal(verts[:, 4], verts[:, 1]) and np.array_equal(verts[:, 5], verts[:, 2]) and (verts[..., 3] == 1.0).all()
    assert np.allclose(verts[:, 3, :3] + verts[:, 0, :3], verts[:, 1, :3] + verts[:, 2, :3], atol=1e-6), "a11 is opposite a00"


def test_lattice_tracer_matches_brute_force(tmp_path):
    """The oracle's particle query (its own BVH over the unit cell + a box test per lattice cell) against numpy's brute force over every triangle of
    every one of the 15625 cells, with the disc cut-out in barycentric space (optix_common.cuh:67-74)."""
    host = _with_particles(scenes.cornell_host(str(tmp_path), 8, 8, 1), count=48, size=60.0, seed=5)
    view = host.device_scene()
    tris = np.ctypeslib.as_array(C.cast(view.particle_vertices, C.POINTER(C.c_float)), shape=(96, 3, 4))[..., :3].astype(np.float64)
    rng = np.random.RandomState(2)
    n = 40
    pos = rng.rand(n, 3).astype(np.float32)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = (d * 0.25).astype(np.float32)   # direction / particles_scale
    tmax = rng.uniform(5.0, 60.0, n).astype(np.float32)
    out_t = np.zeros(n, dtype=np.float32); out_tri = np.zeros(n, dtype=np.uint32)
    L.oracle_probe_particle_trace(C.byref(view), C.c_uint32(n), pos.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), tmax.ctypes.data_as(C.c_void_p),
                                  out_t.ctypes.data_as(C.c_void_p), out_tri.ctypes.data_as(C.c_void_p))
    g = np.arange(-12, 13, dtype=np.float64)
    cells = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    p0, e1, e2 = tris[:, 0], tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]
    hits = 0
    for r in range(n):
        o = pos[r].astype(np.float64)[None, None, :] - cells[:, None, :]     # [cell, 1, 3]
        dd = d[r].astype(np.float64)
        h = np.cross(dd, e2)                                                   # [tri, 3]
        a = (e1 * h).sum(axis=1)
        s = o - p0[None, :, :]                                                 # [cell, tri, 3]
        u = (s * h[None]).sum(axis=2) / a
        q = np.cross(s, e1[None])
        v = (q * dd).sum(axis=2) / a
        t = (q * e2[None]).sum(axis=2) / a
        ok = (u >= 0) & (v >= 0) & (u + v <= 1) & (t >= 0) & (t < tmax[r]) & ((u - 0.5) ** 2 + (v - 0.5) ** 2 <= 0.25)
        if ok.any():
            tt = np.where(ok, t, np.inf)
            best = np.unravel_index(np.argmin(tt), tt.shape)
            assert out_tri[r] == best[1], (r, out_tri[r], best)
            assert abs(out_t[r] - tt[best]) < 1e-4 * max(1.0, tt[best])
            hits += 1
        else:
            assert out_tri[r] == 0xFFFFFFFF
    assert hits >= 10


def test_particles_change_only_what_delta_paths_see(tmp_path):
    host = scenes.cornell_host(str(tmp_path), 40, 28, 3)
    base, _, cnt0 = oracle_lib.render(_view(host), 0, 4)
    _with_particles(host, count=512, size=3.0, scale=2.0)
    fm, _, cnt1 = oracle_lib.render(_view(host), 0, 4)
    assert np.isfinite(fm).all()
    changed = (fm.reshape(3, -1) != base.reshape(3, -1)).any(axis=0).mean()
    assert 0.005 < changed < 0.9, changed   # sparse particles: some pixels hit one, most do not
    assert cnt1[0] != cnt0[0]


# ---------------------------------------------------------------- GPU: HIP == oracle

def _parity(host, samples=3, spp_pass=2, counters=4):
    from luminary_amd.core import Core
    view = _view(host)
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
def test_particles_in_the_zoo_match_the_oracle(mode):
    """Dense, large particles in front of the material zoo (emissive triangles: light sampling from a particle; the sun and the ambient sample in the other
    sky modes), moving along their direction during the exposure."""
    host = scenes.zoo_scene(64, 40, 5, sky_mode=mode)
    _with_particles(host, count=3000, size=20.0, scale=5.0, speed=0.7)
    fm = _parity(host, samples=3)
    plain = scenes.zoo_scene(64, 40, 5, sky_mode=mode)
    assert not np.array_equal(fm, oracle_lib.render(_view(plain), 0, 3)[0])


@pytest.mark.gpu
def test_particles_in_fog_match_the_oracle(tmp_path):
    """Particles and fog together: the scattering-event kernel keeps probability 1 on a delta path that ends on a particle (volume.cuh:150-155), light
    samples of a particle carry the fog's transmittance."""
    host = scenes.cornell_host(str(tmp_path), 48, 32, 4)
    _with_particles(host, count=800, size=4.0, scale=1.5, albedo=(0.6, 0.9, 0.5), diameter=12.0)
    f = host.get_fog(); f.active, f.density = True, 90.0; host.set_fog(f)
    _parity(host, samples=4)


@pytest.mark.gpu
@pytest.mark.parametrize("shading_mode", [1, 2, 3, 4])
def test_particle_debug_modes_match_the_oracle(shading_mode):
    host = scenes.zoo_scene(64, 40, 3)
    _with_particles(host, count=2000, size=25.0, scale=5.0)
    st = host.get_settings(); st.shading_mode = shading_mode; host.set_settings(st)
    _parity(host, samples=2, counters=3)  # the debug kernels do not count vertices


@pytest.mark.gpu
def test_default_sized_particles_and_an_empty_scene():
    """The reference's defaults (8192 particles of size 1 in cells of 10 units) in a scene with no geometry: every camera ray runs its full length through
    the lattice."""
    host = scenes.edge_scene("empty", 64, 40, 2)
    p = host.get_particles(); p.active = True; host.set_particles(p)
    _parity(host, samples=4, spp_pass=4)
