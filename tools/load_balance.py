#!/usr/bin/env python3
"""Load balance of the image-tile deal, measured on ONE GPU: each of the 8 shares of a frame is rendered alone - the work one
rank of an 8-GPU run does - and its rays and milliseconds are recorded. An 8-GPU render ends when its slowest rank ends, so the deal's contribution to the
scaling efficiency is mean(ms) / max(ms). Both deals are measured on the same box in the same process:
  lattice   tile (x, y) -> rank (x + 3 y) % 8                 (round 5, lumc_tile_owner)
  rowmajor  tile t of the row-major grid -> rank t % 8          (rounds 1-4: vertical stripes whenever the tile row length is a multiple of 8)

  python tools/load_balance.py [--scenes hall,scan,example] [--sizes 1920x1080,3840x2160] [--spp 16] [--world 8] > profiles/r05_load_balance.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="hall,scan,example")
    ap.add_argument("--sizes", default="1920x1080,3840x2160")
    ap.add_argument("--spp", type=int, default=16, help="sample ids per share and timing (one pass)")
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--world", type=int, default=8)
    args = ap.parse_args()
    import bench
    from luminary_amd.core import CNT_LIGHT_BVH, CNT_SHADOW, CNT_TRACE, Core
    from luminary_amd.distributed import tile_lattice_step, tile_pixels
    world = args.world
    core = Core(0)
    rows = []
    for name in args.scenes.split(","):
        host = None
        for size in args.sizes.split(","):
            w, h = (int(v) for v in size.split("x"))
            if host is None:
                host = bench.build_workload(name, w, h, 8)
            s = host.get_settings()
            s.width, s.height = w, h
            host.set_settings(s)
            view = host.device_scene()
            core.upload(view)
            for deal in ("lattice", "rowmajor"):
                if deal == "rowmajor":
                    os.environ["LUM_TILE_DEAL"] = "rowmajor"
                else:
                    os.environ.pop("LUM_TILE_DEAL", None)
                shares = [tile_pixels(w, h, r, world) for r in range(world)]
                os.environ.pop("LUM_TILE_DEAL", None)
                assert sum(s.size for s in shares) == w * h
                ms, rays = [], []
                for r in range(world):
                    core.set_pixels(shares[r])
                    core.render(0, args.spp, args.spp)  # warm: work buffers of this share's size
                    core.synchronize()
                    best = 1e30
                    for k in range(args.repeat):
                        core.reset_counters()
                        t0 = time.time()
                        core.render((k + 1) * args.spp, args.spp, args.spp)
                        core.synchronize()
                        best = min(best, (time.time() - t0) * 1e3)
                    c = core.counters()
                    ms.append(best)
                    rays.append(c[CNT_TRACE] + c[CNT_SHADOW] + c[CNT_LIGHT_BVH])
                ms, rays = np.array(ms), np.array(rays, dtype=np.float64)
                row = {"scene": name, "width": w, "height": h, "world": world, "deal": deal, "spp": args.spp,
                       "pixels_per_share": [int(s.size) for s in shares],
                       "ms_per_share": [round(float(x), 3) for x in ms], "rays_per_share": [int(x) for x in rays],
                       "mean_over_max_ms": round(float(ms.mean() / ms.max()), 4), "mean_over_max_rays": round(float(rays.mean() / rays.max()), 4),
                       "min_over_max_ms": round(float(ms.min() / ms.max()), 4)}
                rows.append(row)
                print("%-8s %4dx%-4d %-8s  ms mean/max %.4f  rays mean/max %.4f  (ms %s)" % (name, w, h, deal, row["mean_over_max_ms"], row["mean_over_max_rays"],
                                                                                               " ".join("%.1f" % x for x in ms)), file=sys.stderr, flush=True)
        if host is not None:
            host.close()
    core.close()
    out = {"what": "each of the %d shares of the tile deal rendered alone on one MI355X (fast flavour, 8 bounces, %d spp in one pass, best of %d): predicted multi-GPU "
                   "efficiency of the deal = mean / max of the shares' times" % (world, args.spp, args.repeat),
           "lattice_step": tile_lattice_step(world), "rows": rows}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
