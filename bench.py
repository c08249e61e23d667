#!/usr/bin/env python3
"""Benchmark of the MI355X path-tracing core on BASELINE.json's metric (Mrays/s at 1920x1080, 8 bounces).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

Step   = one wavefront pass per GPU: every rank takes its pixels of the 1920x1080 frame through all 9 depth passes (8 bounces) for
         --samples-per-pass (default 8) x N sample ids, i.e. 1920*1080*8 paths per GPU and step whatever N is (weak scaling: the
         frame gains 8*N samples per step). Deep bounces keep few paths alive, so several sample ids share a pass to keep 256 CUs busy.
Rays   = closest-hit rays + executed shadow rays + light-BVH queries (SURVEY.md §8d counting rule), counted on the device.
N > 1  = the frame is cut into 32x32 tiles dealt round-robin to the ranks (weak data-parallel over pixels, no collective while
         rendering); each rank accumulates its own pixels and one RCCL reduce to rank 0 at the end assembles the frame moments.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
NODE_BYTES, TRI_BYTES = 112, 48  # one node visit fetches 7 x 16 B of a 128-B BVH4 node; triangle = 48 B (DESIGN.md "Algorithmic bytes")
IO_TRACE_BYTES = 24 + 4 + 12  # origin+dir, tmax, hit
IO_SHADOW_BYTES = 24 + 4 + 12  # origin+dir, tmax, RGB visibility
TRAFFIC_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")


def measured_traffic(workload, kernel, spp_per_step):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc pass of this same command (tools/pmc_traffic.py
    writes profiles/pmc_traffic.json; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, plus WRITE_SIZE)."""
    try:
        with open(TRAFFIC_FILE) as f:
            t = json.load(f)
        e = t[workload][kernel]
        if int(e["spp_per_step"]) != int(spp_per_step):
            return None
        return float(e["bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


from luminary_amd.distributed import assemble_frame, tile_pixels  # noqa: E402


def build_workload(name, width, height, bounces):
    from luminary_amd import scenes
    if name == "example":
        return scenes.example_scene(width, height, bounces), "C2 Example-class scene (~100k triangles, 72 instances, 16 emissive quads)"
    if name == "hall":
        return scenes.hall_scene(width, height, bounces), "C3 Sponza-class hall (1M triangles, one mesh, 32 emissive panels)"
    if name == "scan":
        return scenes.scan_scene(width, height, bounces), "C5 scanned-object class (5.2M-triangle displaced icosphere, 8 area lights)"
    if name == "cornell":
        return scenes.cornell_host("/tmp/lum_bench_cornell", width, height, bounces), "C1 Cornell box (36 triangles)"
    raise SystemExit("unknown workload " + name)


def cpu_baseline(view, budget_s):
    """Times the oracle (CPU restatement, OpenMP over pixels) on a bounded sample of the same workload: a strided pixel subset of the
    frame at k spp, sized from a short calibration run so that it takes about `budget_s` seconds. The oracle's own BVH build is timed
    separately (one-pixel call) and subtracted."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    cores = os.cpu_count() or 1
    v = oracle_lib.with_luts(view)  # the energy tables come from the committed fixture
    n_total = view.width * view.height
    t0 = time.time()
    oracle_lib.render(v, 0, 1, pixels=np.array([0], dtype=np.uint32), use_bvh=True, threads=cores)
    t_build = time.time() - t0

    def run(pixels, spp):
        t = time.time()
        _, _, cnt = oracle_lib.render(v, 0, spp, pixels=pixels, use_bvh=True, threads=cores)
        return max(time.time() - t - t_build, 1e-6), float(cnt[0] + cnt[1] + cnt[2])

    # calibrate on a strided subset, then grow the sample until it takes about `budget_s` (at most 3 rounds)
    pixels, spp = np.arange(0, n_total, 97, dtype=np.uint32), 1
    dt, rays = run(pixels, spp)
    for _ in range(3):
        if dt >= 0.5 * budget_s:
            break
        grow = min(max(budget_s / max(dt, 1e-3), 1.5), 64.0)
        npx = n_total if pixels is None else pixels.size
        want_px_spp = npx * spp * grow
        if want_px_spp <= n_total:
            pixels, spp = np.arange(0, n_total, max(1, int(n_total / want_px_spp)), dtype=np.uint32), 1
        else:
            pixels, spp = None, max(1, min(64, int(want_px_spp / n_total)))
        dt, rays = run(pixels, spp)
    npx = n_total if pixels is None else pixels.size
    model = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port", "cpu": model,
            "sample": "oracle/ (CPU restatement, OpenMP over pixels, %d threads) on %d pixels x %d spp of the same frame, all 9 depth passes: "
                      "%.0f rays in %.1f s (its %.1f s BVH build excluded)" % (cores, npx, spp, rays, dt, t_build)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="example")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--samples-per-pass", type=int, default=8, help="sample ids per wavefront pass = per step")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU baseline work (0 = skip)")
    ap.add_argument("--sky", default="constant", choices=["constant", "procedural"],
                    help="constant = the benchmark settings (SURVEY §8d); procedural = sky mode DEFAULT: ray-marched atmosphere and sun sampling")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from luminary_amd.core import Core, CNT_LIGHT_BVH, CNT_NODES, CNT_SHADOW, CNT_TRACE, CNT_TRIS
    host, workload_name = build_workload(args.workload, args.width, args.height, args.bounces)
    if args.sky == "procedural":
        sky = host.get_sky()
        sky.mode = 0
        host.set_sky(sky)
        workload_name += " under the procedural sky (mode DEFAULT)"
    view = host.device_scene()
    core = Core(local_rank)
    t_up = time.time()
    core.upload(view)  # builds the BVHs, generates the BSDF tables on the GPU
    upload_s = time.time() - t_up
    pixels = tile_pixels(view.width, view.height, rank, world) if world > 1 else None
    core.set_pixels(pixels)
    P = core.num_pixels
    fm = torch.zeros(3 * P, dtype=torch.float32, device="cuda")
    sm = torch.zeros(P, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    spp_step = args.samples_per_pass * world  # per-GPU paths per step stay W*H*samples_per_pass

    def step(i):
        core.render(i * spp_step, spp_step, spp_step, fm.data_ptr(), sm.data_ptr(), stream)

    for i in range(args.warmup):
        step(i)
    if dist is not None:
        assemble_frame(fm, sm, pixels, view.width * view.height, dist, 0)  # untimed: creates the RCCL communicator and its buffers
    torch.cuda.synchronize()
    core.synchronize()
    core.reset_counters()
    core.set_profiling(True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(args.steps):
        step(args.warmup + i)
    if dist is not None:
        # assemble the frame on rank 0: every rank scatters its pixels into a zero frame, one reduce over xGMI
        assemble_frame(fm, sm, pixels, view.width * view.height, dist, 0)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.time() - t0

    cnt = core.counters()
    times = core.kernel_times()
    # output chain (tone map + ARGB8 of the frame just rendered), outside the timed region: streaming kernels, 40 B/pixel algorithmic
    output_chain = None
    if world == 1:
        from luminary_amd.core import default_output_params
        op = default_output_params(view.width, view.height, max(spp_step * (args.steps + args.warmup), 1))
        for _ in range(5):
            core.generate_output(op, fm.data_ptr())
        out_ms, out_n = core.kernel_times()["output"]
        if out_n:
            per = out_ms / out_n
            output_chain = {"ms_per_frame": round(per, 4), "GB/s": round(40.0 * view.width * view.height / (per * 1e-3) / 1e9, 1),
                            "frac_of_hbm_peak": round(40.0 * view.width * view.height / (per * 1e-3) / 1e9 / HBM_PEAK_GBPS, 3)}
    rays_local = cnt[CNT_TRACE] + cnt[CNT_SHADOW] + cnt[CNT_LIGHT_BVH]
    stats = torch.tensor([float(rays_local), float(cnt[CNT_TRACE]), float(cnt[CNT_SHADOW]), float(cnt[CNT_LIGHT_BVH]), elapsed], dtype=torch.float64,
                         device="cuda")
    if dist is not None:
        mx = stats.clone()
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        elapsed = float(mx[4])
    rays_total = float(stats[0])
    if rank != 0:
        return

    # roofline of the dominant traversal kernel on rank 0: algorithmic bytes = nodes*128 + triangles*48 + per-ray I/O
    trace_ms, trace_n = times["trace"]
    shadow_ms, shadow_n = times["shadow"]
    nodes_trace, tris_trace, nodes_shadow, tris_shadow = cnt[CNT_NODES], cnt[CNT_TRIS], cnt[6], cnt[7]
    bytes_trace = nodes_trace * NODE_BYTES + tris_trace * TRI_BYTES + cnt[CNT_TRACE] * IO_TRACE_BYTES
    bytes_shadow = nodes_shadow * NODE_BYTES + tris_shadow * TRI_BYTES + cnt[CNT_SHADOW] * IO_SHADOW_BYTES
    dominant = "trace" if trace_ms >= shadow_ms else "shadow"
    dom_bytes, dom_ms, dom_n = (bytes_trace, trace_ms, trace_n) if dominant == "trace" else (bytes_shadow, shadow_ms, shadow_n)
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": "k_" + dominant, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": measured_traffic(args.workload, "k_trace" if dominant == "trace" else "k_shadow_rays", args.samples_per_pass) if world == 1 else None, "avg_launch_ms": dom_ms / max(dom_n, 1), "launches": dom_n,
                "algorithmic_bytes_per_launch": dom_bytes / max(dom_n, 1)}
    # reported at N=1 only; the oracle needs the sky tables for the procedural sky, which the bench does not generate on the CPU
    cpu = cpu_baseline(view, args.cpu_budget) if (args.cpu_budget > 0 and world == 1 and args.sky == "constant") else None
    out = {
        "metric": "Mrays/s at 1920x1080, 8 bounces", "value": rays_total / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": workload_name, "width": view.width, "height": view.height, "max_ray_depth": view.max_ray_depth,
                   "spp_per_step": spp_step, "paths_per_gpu_per_step": P * spp_step, "partition": "32x32 image tiles round-robin over ranks" if world > 1 else "single GPU",
                   "samples_per_s": view.width * view.height * spp_step * args.steps / elapsed,
                   "rays": {"closest": float(stats[1]), "shadow": float(stats[2]), "light_bvh": float(stats[3])},
                   "per_ray_rank0": {"nodes_closest": round(nodes_trace / max(cnt[CNT_TRACE], 1), 2), "tris_closest": round(tris_trace / max(cnt[CNT_TRACE], 1), 2),
                                     "nodes_shadow": round(nodes_shadow / max(cnt[CNT_SHADOW], 1), 2), "tris_shadow": round(tris_shadow / max(cnt[CNT_SHADOW], 1), 2),
                                     "lds_hit_rate_closest": round(cnt[10] / max(nodes_trace, 1), 3), "lds_hit_rate_shadow": round(cnt[11] / max(nodes_shadow, 1), 3)},
                   "kernel_ms_rank0": {k: round(v[0], 3) for k, v in times.items()}, "output_chain_rank0": output_chain, "scene_upload_s": round(upload_s, 2)},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
