#!/usr/bin/env python3
"""Throughput of adaptive rendering (lumc_adaptive_*) next to uniform rendering on a bench workload.
  python tools/adaptive_bench.py [example|hall] [update_interval] [avg_rate]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from luminary_amd.core import Core, default_output_params  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "example"
interval = int(sys.argv[2]) if len(sys.argv) > 2 else 8
avg_rate = int(sys.argv[3]) if len(sys.argv) > 3 else 2
host, label = bench.build_workload(name, 1920, 1080, 8)
core = Core(0)
core.upload(host.device_scene())
core.set_pixels(None)
core.render(0, 8, samples_per_pass=8)
core.synchronize()


def rays():
    c = core.counters()
    return c[0] + c[1] + c[2]


core.reset_counters()
t = time.time()
core.render(8, 16, samples_per_pass=8)
core.synchronize()
dt = time.time() - t
print("%s\nuniform: 16 spp in %.1f ms, %.0f Mrays/s" % (label, dt * 1e3, rays() / dt / 1e6))
tone = default_output_params(1920, 1080, 1)
core.adaptive_begin(256, avg_rate, interval, exposure=1.0, tone=tone)
for stage in range(4):
    n = interval << stage
    core.reset_counters()
    t = time.time()
    core.adaptive_render(n)
    core.synchronize()
    dt = time.time() - t
    info = core.adaptive_info()
    print("stage %d: %d executions in %.1f ms (incl. the build of stage %d), %.0f Mrays/s; next stage: %d tasks per execution, variance total %.4g"
          % (stage, n, dt * 1e3, stage + 1, rays() / dt / 1e6, info["tasks_per_execution"], info["variance_total"]))
