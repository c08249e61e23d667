#!/usr/bin/env python3
"""Build time and trace speed of the two BVH builders on a bench workload: python tools/lbvh_bench.py [example|hall|scan]"""
import os
import sys
import time

os.environ["LUM_BVH_SHARE"] = "0"  # time every build: no tree taken from another context of the process
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from luminary_amd.core import Core  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "hall"
host, label = bench.build_workload(name, 1920, 1080, 8), bench.WORKLOADS[name]
view = host.device_scene()
print(label)
for builder in ("sah", "lbvh", "ploc", "sah_gpu", "sah_gpu"):
    core = Core(0)
    core.set_bvh_builder(builder)
    t = time.time()
    core.upload(view)
    up = time.time() - t
    core.set_pixels(None)
    core.render(0, 8, samples_per_pass=8)
    core.synchronize()
    core.reset_counters()
    t = time.time()
    core.render(8, 16, samples_per_pass=8)
    core.synchronize()
    dt = time.time() - t
    c = core.counters()
    st = core.bvh_stats()
    print("%-4s upload %.2f s (mesh BVH builds %.3f s, %s), %d nodes; %.0f Mrays/s; nodes/tris per closest ray %.1f/%.1f, per shadow ray %.1f/%.1f"
          % (builder, up, core.bvh_build_seconds(), core.bvh_meshes_by_builder(), st[0], (c[0] + c[1] + c[2]) / dt / 1e6, c[4] / max(c[0], 1), c[5] / max(c[0], 1),
             c[6] / max(c[1], 1), c[7] / max(c[1], 1)))
    core.close()
