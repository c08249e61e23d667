#!/usr/bin/env python3
"""fast vs exact flavour on a parity scene: relative L2 of the radiance at growing sample counts (noise from flipped discrete decisions
falls like 1/sqrt(spp), a bias does not), the signed relative difference of the image sums, and where the difference sits.
  python tools/flavour_diff.py [zoo|cornell|textured|hall] [spp ...] [noise]
noise: also renders the exact flavour with the NEXT spp sample ids - the relative L2 between two independent exact estimates is the Monte-Carlo noise
the flavours' difference has to be read against.
hall = the north-star scene at the north-star size (1.43 M triangles, 1920x1080, 8 bounces); the last line printed is a JSON record."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
from luminary_amd import scenes
from luminary_amd.core import Core

name = sys.argv[1] if len(sys.argv) > 1 else "zoo"
with_noise = "noise" in sys.argv[2:]
spps = [int(x) for x in sys.argv[2:] if x != "noise"] or [64, 256, 1024, 4096]
host = {"zoo": lambda: scenes.zoo_scene(96, 64, 8), "cornell": lambda: scenes.cornell_host("/tmp/fd_cornell", 64, 64, 8), "textured": lambda: scenes.textured_scene(96, 64, 6),
        "hall": lambda: scenes.hall_scene(1920, 1080, 8)}[name]()
view = oracle_lib.with_luts(host.device_scene())
core = Core(0)
core.upload(view)
w, h = view.width, view.height
import json
records = []
for spp in spps:
    out = {}
    for fl in ("exact", "fast"):
        core.set_flavour(fl); core.set_pixels(None); core.reset_counters()
        core.render(0, spp, samples_per_pass=min(spp, 32 if name == "hall" else 64))
        fm, _ = core.accumulators()
        out[fl] = (fm.astype(np.float64) / spp, core.counters()[:4])
    e, f = out["exact"][0], out["fast"][0]
    noise = None
    if with_noise and 2 * spp <= (1 << 20):
        core.set_flavour("exact"); core.set_pixels(None)
        core.render(spp, spp, samples_per_pass=min(spp, 32 if name == "hall" else 64))
        other = core.accumulators()[0].astype(np.float64) / spp
        noise = float(np.sqrt(((other - e) ** 2).sum() / (e ** 2).sum()))
    rel = np.sqrt(((f - e) ** 2).sum() / (e ** 2).sum())
    bias = (f.sum() - e.sum()) / e.sum()
    records.append({"scene": name, "spp": spp, "rel_l2": float(rel), "sum_bias": float(bias), "counters_exact": [int(x) for x in out["exact"][1]], "counters_fast": [int(x) for x in out["fast"][1]]})
    d = np.abs(f - e).sum(axis=0).reshape(h, w)
    rows = d.reshape(4, h // 4, 4, w // 4).sum(axis=(1, 3))
    if noise is not None:
        records[-1]["exact_vs_exact_next_ids_rel_l2"] = noise
        print("spp %5d two independent exact estimates differ by rel-L2 %.3e (Monte-Carlo noise): the flavours' difference is %.1f %% of it" % (spp, noise, 100.0 * rel / noise))
    print("spp %5d rel-L2 %.3e  sum bias %+.3e  counters exact %s fast %s  differing pixels %d/%d" % (spp, rel, bias, out["exact"][1], out["fast"][1], int((d > 0).sum()), w * h))
    print("   |diff| by 4x4 image regions (share):", np.array2string(rows / max(rows.sum(), 1e-30), precision=2, suppress_small=True).replace("\n", " "))
print(json.dumps(records))
