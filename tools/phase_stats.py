#!/usr/bin/env python3
"""Where do the lanes of the persistent ray kernels idle? Needs a diagnostic build:
  LUM_CXXFLAGS=-DLUM_PHASE_STATS python -m luminary_amd.build --force && python tools/phase_stats.py [example|hall|scan]
Prints wave-level iteration counts per traversal phase and the lane occupancy of each (both ray kernels together)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import luminary_amd  # noqa: E402
from luminary_amd.core import Core  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "example"
host, label = bench.build_workload(name, 1920, 1080, 8), bench.WORKLOADS[name]
core = Core(0)
core.upload(host.device_scene())
core.set_pixels(None)
lib = luminary_amd._lib()
stats_fn = lib.lumc_debug_phase_stats_fast if core.flavour == "fast" else lib.lumc_debug_phase_stats  # one counter block per flavour
out = (C.c_uint64 * 16)()
core.render(0, 32, samples_per_pass=32)
stats_fn(out, 1)
core.reset_counters()
core.render(32, 32, samples_per_pass=32)
stats_fn(out, 1)
cnt = core.counters()
node_it, inst_it, inst_l, tri_it, tri_l, outer, pop_it, pop_l = [int(x) for x in out][:8]
sh = [int(x) for x in out][8:]
nodes = int(cnt[4] + cnt[6]); tris = int(cnt[5] + cnt[7]); rays = int(cnt[0] + cnt[1])
print(label, "- flavour", core.flavour)
print("rays %d  node visits %d  triangle tests %d" % (rays, nodes, tris))
print("node phase:     %10d wave iterations, lane occupancy %.3f" % (node_it, nodes / (64.0 * max(node_it, 1))))
print("pop:            %10d wave iterations, lane occupancy %.3f" % (pop_it, pop_l / (64.0 * max(pop_it, 1))))
print("instance entry: %10d wave iterations, lane occupancy %.3f" % (inst_it, inst_l / (64.0 * max(inst_it, 1))))
print("triangle phase: %10d wave iterations, lane occupancy %.3f, triangle slots used %.3f of %d" % (tri_it, tri_l / (64.0 * max(tri_it, 1)), tris / max(tri_l, 1), int(lib.lumc_leaf_max_triangles())))
print("outer iterations (refill checks): %d" % outer)
print("per ray: %.2f node, %.2f instance, %.2f leaf visits" % (nodes / rays, inst_l / rays, tri_l / rays))
if core.flavour == "fast" and hasattr(lib, "lumc_debug_phase_times_fast"):
    tm = (C.c_uint64 * 16)()
    lib.lumc_debug_phase_times_fast(tm, 1)
    tm = [int(x) for x in tm]
    # s_memtime counts at 100 MHz on this part (constant clock), so one tick = 10 ns
    names = ["node iterations that touch memory", "node iterations on staged nodes only", "triangle iterations", "instance-entry iterations", "refills", "whole kernel per wave"]
    for k, nm in enumerate(names):
        c, n = tm[2 * k], tm[2 * k + 1]
        if n:
            print("time: %-40s %12d  avg %8.1f ticks  total %14d ticks" % (nm, n, c / n, c))
print("shade: light sampling entered by %d waves, lane occupancy %.3f" % (sh[6], sh[7] / (64.0 * max(sh[6], 1))))
print("shade: candidate loop %d wave iterations, occupancy %.3f; BSDF+MIS part %d wave iterations, occupancy %.3f (%.2f of 8 candidates per vertex)"
      % (sh[0], sh[1] / (64.0 * max(sh[0], 1)), sh[2], sh[3] / (64.0 * max(sh[2], 1)), sh[3] / max(sh[7], 1)))
print("shade: light-tree descent %d wave iterations, occupancy %.3f" % (sh[4], sh[5] / (64.0 * max(sh[4], 1))))
if core.flavour == "fast" and hasattr(lib, "lumc_debug_shade_times_fast"):
    tm = (C.c_uint64 * 16)()
    lib.lumc_debug_shade_times_fast(tm, 1)
    tm = [int(x) for x in tm]
    names = {0: "queue words, surface context, local frame", 1: "light-tree root pass + energy terms", 10: "candidate: pick + triangle sample", 11: "candidate: colour + BSDF",
             12: "candidate: MIS + reservoir", 2: "candidate loop remainder", 3: "BSDF-driven light direction", 4: "bounce + ambient record", 5: "sun", 6: "NEE stores, classification, roulette",
             7: "appends", 8: "collecting hits (input rounds)"}
    total = sum(tm[k] for k in names)
    print("shade time (s_memtime ticks per wave, %d batches of 64 vertices; shares of %d ticks):" % (tm[9], total))
    for k, nm in names.items():
        print("  %-45s %6.1f %%   %8.1f ticks per batch" % (nm, 100.0 * tm[k] / max(total, 1), tm[k] / max(tm[9], 1)))
if core.flavour == "fast" and hasattr(lib, "lumc_debug_vis_stats_fast"):
    vs = (C.c_uint64 * 8)()
    lib.lumc_debug_vis_stats_fast(vs, 1)
    vs = [int(x) for x in vs]
    for k, nm in enumerate(["sampled light (segment)", "BSDF-sampled light (segment)", "ambient (no end point)", "sun (no end point)"]):
        if vs[2 * k]:
            print("visibility rays, %-30s %12d, blocked by an opaque surface %.3f" % (nm, vs[2 * k], vs[2 * k + 1] / vs[2 * k]))
