#!/usr/bin/env python3
"""Locates where the HIP path and the oracle part ways on the zoo scene (run on the GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from luminary_amd import scenes
from luminary_amd.core import Core
import oracle_lib

sky, ap, bl = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
core = Core(0)


def both(bounces, first, count, pixels=None):
    v = oracle_lib.with_luts(scenes.zoo_scene(96, 64, bounces, sky_mode=sky, aperture=ap, blades=bl).device_scene())
    core.upload(v)
    core.set_pixels(pixels)
    core.reset_counters()
    core.render(first, count, samples_per_pass=count)
    fm, sm = core.accumulators()
    ofm, osm, ocnt = oracle_lib.render(v, first, count, pixels=pixels)
    return fm, ofm, core.counters()[:4], [int(x) for x in ocnt[:4]]


fm, ofm, c, oc = both(8, 5, 3)
bad = np.unique(np.argwhere(fm != ofm)[:, 1])
print("differing pixels", bad.tolist(), "counters", c, oc)
for p in bad[:4]:
    px = np.array([p], dtype=np.uint32)
    for s in (5, 6, 7):
        for b in range(0, 9):
            f, o, c, oc = both(b, s, 1, px)
            if not np.array_equal(f, o) or c != oc:
                print("pixel", int(p), "sample", s, "first differs at max depth", b, f.ravel().tolist(), o.ravel().tolist(), c, oc,
                      "bits", [hex(int(x)) for x in f.ravel().view(np.uint32)], [hex(int(x)) for x in o.ravel().view(np.uint32)])
                break
