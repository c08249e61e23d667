"""GPU builders (lbvh.hip: the Morton-ordered radix tree, and parallel locally-ordered clustering over the same order): the trees they build give
the same hits and the same image as the oracle (and therefore as the host SAH builder): results never depend on the acceleration structure."""
import ctypes

import numpy as np
import pytest

import oracle_lib
from luminary_amd import scenes
from luminary_amd.core import Core

pytestmark = pytest.mark.gpu


def _rays(n, seed, lo, hi):
    rng = np.random.RandomState(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


@pytest.mark.parametrize("builder", ["lbvh", "ploc", "sah_gpu"])
def test_lbvh_trees_trace_and_render_like_the_oracle(builder):
    host = scenes.example_scene(96, 54, 6, sphere_segments=10, ground_res=24, num_objects=24, num_lights=6)
    view = oracle_lib.with_luts(host.device_scene())
    core = Core(0)
    try:
        core.set_bvh_builder(builder)
        core.upload(view)
        used = core.bvh_meshes_by_builder()
        assert used["lbvh"] >= 2 and used["sah"] <= 1, used  # a one-triangle mesh has nothing to sort and takes the host path
        stats_lbvh = core.bvh_stats()
        o, d = _rays(40000, 5, -30.0, 30.0)
        o[:, 1] = np.abs(o[:, 1]) * 0.5 + 0.5
        ign = np.full((o.shape[0], 2), 0xFFFFFFFF, dtype=np.uint32)
        got = core.trace_closest_host(o, d, ign)
        want = oracle_lib.trace_closest(view, o, d, ign, use_bvh=True)
        assert np.array_equal(got, want), "closest hits on LBVH trees"
        core.set_pixels(None)
        core.render(0, 2, samples_per_pass=2)
        fm, sm = core.accumulators()
        ofm, osm, _ = oracle_lib.render(view, 0, 2)
        assert np.array_equal(fm, ofm) and np.array_equal(sm, osm), "image on LBVH trees"
        # same scene through the default builder: different trees (node counts differ), same image
        core.set_bvh_builder("sah")
        core.upload(view)
        assert core.bvh_meshes_by_builder()["lbvh"] == 0
        assert core.bvh_stats()[1] == stats_lbvh[1], "same triangles"
        core.set_pixels(None)
        core.render(0, 2, samples_per_pass=2)
        fm2, _ = core.accumulators()
        assert np.array_equal(fm2, ofm)
    finally:
        core.close()


@pytest.mark.parametrize("builder", ["lbvh", "ploc", "sah_gpu"])
@pytest.mark.parametrize("kind", ["degenerate", "one_triangle", "empty"])
def test_lbvh_edge_scenes(kind, builder):
    """Zero-area and duplicated triangles (equal Morton codes: ties are broken by the index bits), a single triangle, no geometry."""
    host = scenes.edge_scene(kind, 48, 32, 4)
    view = oracle_lib.with_luts(host.device_scene())
    core = Core(0)
    try:
        core.set_bvh_builder(builder)
        core.upload(view)
        core.set_pixels(None)
        core.render(0, 3, samples_per_pass=2)
        fm, sm = core.accumulators()
        ofm, osm, _ = oracle_lib.render(view, 0, 3)
        assert np.array_equal(fm, ofm) and np.array_equal(sm, osm)
    finally:
        core.close()


def test_the_clustered_trees_are_better_than_the_radix_trees():
    """What the second GPU builder is for: on a mesh with structure (the hall) closest-hit rays visit fewer nodes in its trees than in the radix
    tree's (tools/bvh_quality.cpp models both on the CPU: 19.3 against 22.1, the SAH builder's 17.5)."""
    host = scenes.hall_scene(320, 180, 4, target_triangles=200_000)
    view = host.device_scene()
    visits = {}
    for builder in ("lbvh", "ploc", "sah"):
        core = Core(0)
        try:
            core.set_bvh_builder(builder)
            core.upload(view)
            core.set_pixels(None)
            core.reset_counters()
            core.render(0, 2, samples_per_pass=2)
            cnt = core.counters()
            visits[builder] = cnt[4] / max(cnt[0], 1)
        finally:
            core.close()
    assert visits["ploc"] < visits["lbvh"], visits
    assert visits["sah"] <= visits["ploc"] * 1.02, visits


def test_the_gpu_sah_builder_builds_the_host_builders_trees():
    """lbvh.hip build_bvh4_sah_gpu is the host builder's binned SAH done level by level on the device with the host's own expressions: on meshes without
    degenerate sets (the hall, the scan-class sphere) the trees are the same trees - same node count, and every ray visits exactly as many nodes and tests
    exactly as many triangles as in the host builder's tree (closest-hit and visibility rays of a render; 10^8 visits) - at a fraction of the build time."""
    for name, host in (("hall", scenes.hall_scene(320, 180, 4, target_triangles=200_000)), ("scan", scenes.scan_scene(320, 180, 4, level=6))):
        view = host.device_scene()
        seen = {}
        for builder in ("sah", "sah_gpu"):
            core = Core(0)
            try:
                core.set_bvh_builder(builder)
                core.upload(view)
                used = core.bvh_meshes_by_builder()
                if builder == "sah_gpu":
                    assert used["lbvh"] >= 1, (name, used)  # built on the device (no fallback to the host builder)
                core.set_pixels(None)
                core.reset_counters()
                core.render(0, 2, samples_per_pass=2)
                cnt = core.counters()
                fm, _ = core.accumulators()
                seen[builder] = (core.bvh_stats()[0], cnt[:8], fm.copy(), core.bvh_build_seconds())
            finally:
                core.close()
        assert seen["sah"][0] == seen["sah_gpu"][0], "%s: node counts %d vs %d" % (name, seen["sah"][0], seen["sah_gpu"][0])
        assert seen["sah"][1] == seen["sah_gpu"][1], "%s: rays, node visits and triangle tests %s vs %s" % (name, seen["sah"][1], seen["sah_gpu"][1])
        assert np.array_equal(seen["sah"][2], seen["sah_gpu"][2])


@pytest.mark.parametrize("builder", ["sah_gpu", "ploc"])
def test_gpu_builders_on_triangle_soups(builder):
    """Random triangle soups the level-synchronous builders have to get through: a handful of triangles, many exact duplicates (sets whose centroids all coincide:
    no axis separates them, the GPU SAH builder halves such a set by position), tiny and huge coordinates, long thin triangles. Closest hits against the
    oracle's brute force (no tree at all)."""
    from luminary_amd import Host
    rng = np.random.RandomState(11)
    cases = []
    for n, scale, dup in ((5, 1.0, 0), (17, 1e-3, 0), (300, 1e3, 0), (4000, 1.0, 0), (600, 1.0, 40), (64, 1.0, 64)):
        c = rng.uniform(-1.0, 1.0, (n, 1, 3)) * 10.0
        t = c + rng.normal(size=(n, 3, 3)) * rng.choice([0.05, 0.5, 4.0], size=(n, 1, 1))
        if dup:
            t[-dup:] = t[0]  # `dup` exact copies of one triangle
        cases.append((t * scale).astype(np.float32))
    for tris in cases:
        host = Host()
        scenes.apply_benchmark_settings(host, 16, 16, 2, sky=(0.5, 0.5, 0.5))
        mat = host.add_material(scenes._material((0.6, 0.6, 0.6), 0.6))
        host.new_instance(host.add_mesh(tris.reshape(len(tris), 9), np.full(len(tris), mat, dtype=np.uint16)))
        scenes.set_camera(host, (0.0, 0.0, 30.0), (0.0, 0.0, 0.0))
        view = oracle_lib.with_luts(host.device_scene())
        span = float(np.abs(tris).max())
        o, d = _rays(20000, 3, -1.5 * span, 1.5 * span)
        ign = np.full((o.shape[0], 2), 0xFFFFFFFF, dtype=np.uint32)
        core = Core(0)
        try:
            core.set_bvh_builder(builder)
            core.upload(view)
            got = core.trace_closest_host(o, d, ign)
        finally:
            core.close()
        want = oracle_lib.trace_closest(view, o, d, ign, use_bvh=False)
        assert np.array_equal(got, want), "%s: %d triangles, extent %g" % (builder, len(tris), span)
        host.close()


def test_the_contexts_of_a_process_share_a_meshs_tree(monkeypatch):
    """A host with several devices uploads the same meshes to every context: the first context builds a mesh's tree, the others take it (core.hip
    find_mesh_tree: keyed by the triangles' words, the builder and the builders' environment) - same nodes, same image, a fraction of the time; a mesh
    that differs in one coordinate is built, and with the last context that holds it a tree is released (the next upload builds again)."""
    monkeypatch.delenv("LUM_BVH_SHARE", raising=False)
    view = scenes.hall_scene(160, 90, 3, target_triangles=200_000).device_scene()

    def upload(v):
        core = Core(0)
        core.set_bvh_builder("sah")  # the slowest builder: the difference between building and taking is the largest
        core.upload(v)
        core.set_pixels(None)
        core.render(0, 2, samples_per_pass=2)
        return core, core.bvh_build_seconds(), core.bvh_stats()[0], core.accumulators()[0].copy(), core.bvh_meshes_by_builder()

    first, t_first, nodes_first, frame_first, by_first = upload(view)
    second, t_second, nodes_second, frame_second, by_second = upload(view)
    assert nodes_second == nodes_first and np.array_equal(frame_second, frame_first) and by_second == by_first
    assert t_second < 0.25 * t_first, "the second context built its own tree: %.3f s after %.3f s" % (t_second, t_first)
    moved = scenes.hall_scene(160, 90, 3, target_triangles=200_000).device_scene()
    ctypes.cast(moved.vertices, ctypes.POINTER(ctypes.c_float))[5] += 0.25  # one coordinate of one triangle: another mesh
    third, t_third, _, _, _ = upload(moved)
    assert t_third > 4.0 * t_second, "a different mesh was served from the cache (%.3f s; a taken tree costs %.3f s, a build %.3f s)" % (t_third, t_second, t_first)
    for c in (first, second, third):
        c.close()
    again, t_again, nodes_again, frame_again, _ = upload(view)
    assert t_again > 4.0 * t_second, "the tree outlived its contexts (%.3f s; a taken tree costs %.3f s, a build %.3f s)" % (t_again, t_second, t_first)  # (the first build also warms the builder's threads: builds differ by 2 x)
    assert nodes_again == nodes_first and np.array_equal(frame_again, frame_first)
    again.close()
