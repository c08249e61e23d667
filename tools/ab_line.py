#!/usr/bin/env python3
"""One-line summary of a bench.py JSON line read from stdin (used by tools/gpu_ab.sh and tools/gpu_ab_env.sh)."""
import json, sys
try:
    d = json.loads(sys.stdin.read().strip().split("\n")[-1])
except Exception as e:  # noqa: BLE001
    print("no bench line (%s)" % e)
    sys.exit(0)
if "kernel_ms_rank0" not in d["config"] and d.get("detail"):  # the headline is short since round 4: kernel times and per-ray counts sit in the sidecar the same run wrote
    import os
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), d["detail"])) as f:
        d = json.load(f)
k, p = d["config"]["kernel_ms_rank0"], d["config"]["per_ray_rank0"]
print("%.1f Mrays/s %.4g samples/s trace %.1f shade %.1f shadow %.1f sort %.1f lq %.1f res %.1f vol %.1f | nodes %.2f/%.2f tris %.2f/%.2f lds %.3f/%.3f upload %.2fs" % (
    d["value"], d["config"].get("samples_per_s", 0.0), k["trace"], k["shade"], k["shadow"], k.get("sort", 0.0), k["light_query"], k["resolve"], k.get("volume", 0.0), p["nodes_closest"], p["nodes_shadow"], p["tris_closest"],
    p["tris_shadow"], p.get("lds_hit_rate_closest", 0.0), p.get("lds_hit_rate_shadow", 0.0), d["config"]["scene_upload_s"]))
