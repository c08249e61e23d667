#!/usr/bin/env python3
"""What ONE rank of an N-GPU bench run does, timed on one GPU against the single-GPU step (round 6): bench.py at N ranks gives every rank its tile share of the
frame (1 / N of the pixels, lumc_tile_pixels) and samples-per-pass x N sample ids per step, so that the paths per rank and step are those of the single-GPU run.
If a share with N x the ids ran slower than the whole frame with 1 x - less coherent camera rays, a larger Sobol table, other queue orders - the scaling curve would
show it as lost efficiency before any communication is involved. Output: rate of the whole frame (pixel-samples per second), rate of each emulated rank, their ratio;
predicted weak-scaling efficiency without communication = min over ranks of that ratio (the step ends with the slowest rank).

  python tools/rank_emulation.py [--scenes hall,scan,example] [--world 8] [--spp 64] [--steps 2] > profiles/r06_rank_emulation.json"""
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
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--spp", type=int, default=64, help="sample ids per pass of the single-GPU run (bench.py's default)")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--ranks", default="0,3,7", help="which ranks' shares to time (all of them: 0,1,...)")
    args = ap.parse_args()
    import bench
    from luminary_amd.core import Core
    from luminary_amd.distributed import tile_pixels
    w, h, world = 1920, 1080, args.world
    core = Core(0)
    rows = []

    def rate(pixels, ids):
        core.set_pixels(pixels)
        n = w * h if pixels is None else int(pixels.size)
        core.render(0, ids, ids)  # warm-up: buffers of this size, first touch
        core.synchronize()
        t0 = time.time()
        for k in range(args.steps):
            core.render((k + 1) * ids, ids, ids)
        core.synchronize()
        return n * ids * args.steps / (time.time() - t0)

    for name in args.scenes.split(","):
        host = bench.build_workload(name, w, h, 8)
        core.upload(host.device_scene())
        whole = rate(None, args.spp)
        per_rank = {}
        for r in (int(x) for x in args.ranks.split(",")):
            per_rank[r] = rate(tile_pixels(w, h, r, world), args.spp * world)
        worst = min(per_rank.values())
        rows.append({"scene": name, "world": world, "ids_per_pass_single": args.spp, "ids_per_pass_rank": args.spp * world, "whole_frame_samples_per_s": whole,
                     "rank_samples_per_s": per_rank, "rank_over_whole": {r: v / whole for r, v in per_rank.items()}, "predicted_efficiency_without_communication": worst / whole})
        print("%-8s whole frame %.4g samples/s; ranks %s; worst / whole = %.4f" % (name, whole, " ".join("%d: %.4g" % kv for kv in per_rank.items()), worst / whole), file=sys.stderr, flush=True)
        host.close()
    core.close()
    print(json.dumps({"what": "one rank's work of an N-GPU bench step (1/N of the pixels by the tile deal, N x the sample ids per pass) against the single-GPU step, on one MI355X, fast flavour, 1920x1080, 8 bounces",
                      "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
