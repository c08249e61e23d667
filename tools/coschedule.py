#!/usr/bin/env python3
"""Co-scheduling experiment: k_shade is bound by vector-instruction issue (wait fraction 0.33), the ray kernels wait for memory 56-61 % of
their time at half-empty waves. Two half-batches on two HIP streams, each with its own context (queues, scene replica), let the hardware run one batch's shading
beside the other's ray kernels wherever a CU has room. Host-side only: the kernels are the product's, nothing is rewritten.

  python tools/coschedule.py [--workload hall] [--spp 32] [--steps 6]

  one      one context, `spp` sample ids per pass                                  (the bench's configuration)
  two      two contexts on two streams and two host threads, spp / 2 ids per pass each: the same number of paths in flight
  two_full two contexts, spp ids per pass each: twice the paths in flight
Prints samples/s of each arrangement (same box, same process, interleaved repeats)."""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="hall")
    ap.add_argument("--spp", type=int, default=32)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--repeats", type=int, default=3)
    args = ap.parse_args()
    import torch
    import bench
    from luminary_amd.core import Core
    host = bench.build_workload(args.workload, 1920, 1080, 8)
    view = host.device_scene()
    cores = [Core(0), Core(0)]
    for c in cores:
        c.upload(view)
        c.set_pixels(None)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    P = view.width * view.height

    def run(core, stream, first, per_pass, steps):
        for i in range(steps):
            core.render(first + i * per_pass, per_pass, per_pass, 0, 0, stream.cuda_stream)
        core.synchronize()

    def timed(arrangement):
        spp, steps = args.spp, args.steps
        if arrangement == "one":
            jobs = [(cores[0], streams[0], 0, spp, steps)]
        elif arrangement == "two":
            jobs = [(cores[0], streams[0], 0, spp // 2, steps), (cores[1], streams[1], 1 << 16, spp // 2, steps)]
        else:
            jobs = [(cores[0], streams[0], 0, spp, steps), (cores[1], streams[1], 1 << 16, spp, steps)]
        total = sum(j[3] * j[4] for j in jobs)
        threads = [threading.Thread(target=run, args=j) for j in jobs]
        torch.cuda.synchronize()
        t0 = time.time()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        torch.cuda.synchronize()
        dt = time.time() - t0
        return P * total / dt

    for a in ("one", "two", "two_full"):
        timed(a)  # warm: work buffers of every size
    out = {a: [] for a in ("one", "two", "two_full")}
    for _ in range(args.repeats):
        for a in out:
            out[a].append(timed(a))
    res = {a: {"samples_per_s": max(v), "all": [round(x / 1e6, 1) for x in v]} for a, v in out.items()}
    res["two_over_one"] = res["two"]["samples_per_s"] / res["one"]["samples_per_s"]
    res["two_full_over_one"] = res["two_full"]["samples_per_s"] / res["one"]["samples_per_s"]
    res["workload"], res["spp_per_pass"], res["steps"] = args.workload, args.spp, args.steps
    print(json.dumps(res, indent=1))
    for c in cores:
        c.close()


if __name__ == "__main__":
    main()
