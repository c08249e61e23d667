"""The collapse of the builders' binary trees into 4-wide nodes (csrc/host/bvh_build.cpp CollapsePlan; round 6). CPU: the host builder through
lumc_host_bvh_probe - the optimal cut against the greedy rule of rounds 1-5 on the same binary tree. The GPU builders' collapse follows the same plan
(lbvh.hip k_plan_pass): tests/test_lbvh.py compares their trees with the host's visit for visit."""
import ctypes as C

import numpy as np
import pytest

import luminary_amd


def _probe(boxes, max_leaf):
    lib = luminary_amd._lib()
    lib.lumc_host_bvh_probe.restype = C.c_int
    lib.lumc_host_bvh_probe.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_double * 3)]
    b = np.ascontiguousarray(boxes, dtype=np.float32)
    out = (C.c_uint64 * 6)()
    area = (C.c_double * 3)()
    rc = lib.lumc_host_bvh_probe(b.ctypes.data, b.shape[0], max_leaf, out, area)
    assert rc == 0
    return {"nodes": out[0], "leaves": out[1], "largest_leaf": out[2], "levels": out[3], "slots": out[4], "sound": out[5], "area": area[0], "plan": area[1], "brute": area[2]}


def _soup(rng, n, kind):
    if kind == "uniform":  # small triangles all over a box
        c = rng.uniform(-10, 10, (n, 3)); e = rng.uniform(0.01, 0.2, (n, 3))
    elif kind == "mixed":  # long thin, large and tiny boxes mixed (what the hall's arcades look like to a builder)
        c = rng.uniform(-10, 10, (n, 3)); e = np.exp(rng.uniform(np.log(1e-3), np.log(3.0), (n, 3)))
    elif kind == "sheet":  # a tessellated surface: every box touches its neighbours
        g = int(np.ceil(np.sqrt(n)))
        ij = np.stack(np.meshgrid(np.arange(g), np.arange(g), indexing="ij"), -1).reshape(-1, 2)[:n].astype(np.float64)
        c = np.concatenate([ij * 0.1, np.sin(ij[:, :1] * 0.05) * 2.0], 1); e = np.full((n, 3), 0.06)
    else:  # duplicates: sets no plane separates (the builder's median fallback)
        c = np.repeat(rng.uniform(-1, 1, (n // 8 + 1, 3)), 8, 0)[:n]; e = np.full((n, 3), 0.05)
    return np.concatenate([c - e, c + e], 1).astype(np.float32)


@pytest.mark.parametrize("kind", ["uniform", "mixed", "sheet", "duplicates"])
@pytest.mark.parametrize("n", [1, 2, 5, 37, 4000, 60000])
def test_the_optimal_cut_is_a_sound_tree_and_never_dearer_than_the_greedy_rule(kind, n, monkeypatch):
    rng = np.random.RandomState(n * 7 + len(kind))
    boxes = _soup(rng, n, kind)
    leaf_max = int(luminary_amd._lib().lumc_leaf_max_triangles())
    monkeypatch.setenv("LUM_BVH_COLLAPSE", "0")
    greedy = _probe(boxes, leaf_max)
    monkeypatch.setenv("LUM_BVH_COLLAPSE", "1")
    best = _probe(boxes, leaf_max)
    for t in (greedy, best):
        assert t["sound"] == 1, "every primitive in exactly one leaf, every child box around what is below it"
        assert 1 <= t["largest_leaf"] <= leaf_max
        assert t["slots"] == t["leaves"] + t["nodes"] - 1, "every node but the root hangs in one slot"
    assert best["leaves"] == greedy["leaves"], "the collapse only chooses which binary nodes survive: the leaf sets are the binary tree's"
    # the padded child boxes enter the probe's sum, so the comparison gets the padding's slack (1e-5 relative per coordinate)
    assert best["area"] <= greedy["area"] * (1.0 + 1e-4), (best, greedy)
    assert best["nodes"] <= greedy["nodes"]
    if n >= 4000 and kind != "duplicates":
        assert best["nodes"] < 0.95 * greedy["nodes"], "the greedy rule leaves slots empty (two-leaf nodes at the bottom): %r vs %r" % (best, greedy)
        assert best["slots"] / best["nodes"] > greedy["slots"] / greedy["nodes"] + 0.1, "fuller nodes: %r vs %r" % (best, greedy)


@pytest.mark.parametrize("kind", ["uniform", "mixed", "sheet", "duplicates"])
def test_the_optimal_cut_against_exhaustion_on_small_trees(kind, monkeypatch):
    """Trees small enough to try everything (LUM_BVH_COLLAPSE_BRUTE: every subset of the binary tree's inner nodes as the surviving set, valid when no survivor has
    more than four nearest survivors-or-leaves below it): none is cheaper than the plan's, and the plan's own number is the cost of the tree it builds."""
    monkeypatch.setenv("LUM_BVH_COLLAPSE", "1")
    monkeypatch.setenv("LUM_BVH_COLLAPSE_BRUTE", "1")
    tried = 0
    for seed in range(40):
        rng = np.random.RandomState(1000 + seed)
        n = int(rng.randint(2, 22))
        boxes = _soup(rng, n, kind)
        for leaf in (1, 2):
            t = _probe(boxes, leaf)
            assert t["sound"] == 1
            if t["brute"] < 0.0:
                continue
            tried += 1
            assert abs(t["plan"] - t["brute"]) <= 1e-6 * max(t["brute"], 1e-30), (seed, n, leaf, t)
            if t["leaves"] > 1:  # (a set that fits one leaf has no binary inner node: the plan is empty, the tree a root with one leaf child)
                assert abs(t["area"] - t["plan"]) <= 2e-3 * max(t["plan"], 1e-30), "the tree that was built costs what the plan said (the stored boxes are padded)"
    assert tried >= 40


@pytest.mark.gpu
@pytest.mark.parametrize("builder", ["sah", "sah_gpu", "lbvh", "ploc"])
def test_the_image_does_not_depend_on_the_collapse_rule(builder, monkeypatch):
    """Greedy rule against optimal cut, through every builder (host: bvh_build.cpp collapse; GPU: lbvh.hip k_plan_pass + k_lbvh_collapse): the trees differ - fewer,
    fuller nodes under the optimal cut - and the image, the closest hits and the oracle's image are the same bits (hits are resolved by (t, instance, triangle),
    never by traversal order). LUM_BVH_SHARE=0: every upload builds its own trees."""
    import oracle_lib
    from luminary_amd import scenes
    from luminary_amd.core import Core
    monkeypatch.setenv("LUM_BVH_SHARE", "0")
    host = scenes.example_scene(96, 54, 6, sphere_segments=12, ground_res=32, num_objects=24, num_lights=6)
    view = oracle_lib.with_luts(host.device_scene())
    ofm, osm, _ = oracle_lib.render(view, 0, 2)
    nodes, frames = {}, {}
    for rule in ("0", "1"):
        monkeypatch.setenv("LUM_BVH_COLLAPSE", rule)
        core = Core(0)
        try:
            core.set_bvh_builder(builder)
            core.upload(view)
            nodes[rule] = core.bvh_stats()[0]
            core.set_pixels(None)
            core.render(0, 2, samples_per_pass=2)
            frames[rule] = core.accumulators()
        finally:
            core.close()
    assert nodes["1"] < nodes["0"], "the optimal cut needs fewer 4-wide nodes: %r" % (nodes,)
    for rule in ("0", "1"):
        assert np.array_equal(frames[rule][0], ofm) and np.array_equal(frames[rule][1], osm), "collapse rule %s, builder %s" % (rule, builder)
