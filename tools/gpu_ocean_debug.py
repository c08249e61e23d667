"""Debug helper (GPU): one ocean configuration through the HIP path and the oracle, with per-pixel difference statistics."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import oracle_lib
from luminary_amd import SKY_MODE_CONSTANT_COLOR, SKY_MODE_DEFAULT, SKY_MODE_HDRI, scenes
from luminary_amd.core import Core
from test_particles import _view


def run(name, host, samples=2, depth=None):
    if depth is not None:
        st = host.get_settings(); st.max_ray_depth = depth; host.set_settings(st)
    view = _view(host)
    core = Core(0)
    core.upload(view)
    core.set_pixels(None)
    core.reset_counters()
    core.render(0, samples, samples_per_pass=int(os.environ.get("SPP", "1")))
    fm, sm = core.accumulators()
    ofm, osm, ocnt = oracle_lib.render(view, 0, samples)
    bad = fm != ofm
    print(name, "depth", depth, "differ %d of %d" % (bad.sum(), fm.size), "max abs %g" % np.abs(fm - ofm).max(), "counters", core.counters()[:4], [int(x) for x in ocnt[:4]],
          flush=True)
    core.close()


def ocean(host, **kw):
    o = host.get_ocean()
    o.active = True
    for k, v in kw.items():
        setattr(o, k, v)
    host.set_ocean(o)
    return host


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "empty"
    for depth in (0, 1, 2, 4):
        if which == "empty":
            host = scenes.edge_scene("empty", 48, 32, 4)
            scenes.set_camera(host, (0.0, 3.0, 0.0), (-0.5, 0.0, 0.0))
            run("empty", ocean(host, height=0.0, amplitude=0.5, frequency=0.5), depth=depth)
        else:
            mode = {"const": SKY_MODE_CONSTANT_COLOR, "default": SKY_MODE_DEFAULT, "hdri": SKY_MODE_HDRI}[which]
            for height in (1.0, 4.5):
                run("zoo %s h=%g" % (which, height), ocean(scenes.zoo_scene(64, 40, 5, sky_mode=mode), height=height, amplitude=0.3, frequency=0.5), depth=depth)
