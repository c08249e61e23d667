#!/usr/bin/env python3
"""Benchmark of the MI355X path-tracing core on BASELINE.json's metric (Mrays/s at 1920x1080, 8 bounces).

  python bench.py --gpus N --steps K --warmup W
  N > 1 without a torchrun environment: bench.py starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
  127.0.0.1 --master-port P bench.py ...` as a CHILD (before anything touches the GPU) and exits with its code.

Headline workload = the scene the north-star target is quoted on: the 1 M-triangle Sponza-class hall (BASELINE config 3). At N = 1 the
Example-class scene (config 2) and the 10 M-triangle scan (config 5's scene) are timed after it and reported under "secondary" in the
same JSON line (`--secondary none` skips them).

Step   = one wavefront pass per GPU: every rank takes its pixels of the 1920x1080 frame through all 9 depth passes (8 bounces) for
         --samples-per-pass (default 64) x N sample ids, i.e. 1920*1080*64 = 133 M paths (66 GB of queues, under a quarter of the 288 GB) per GPU and
         step whatever N is (weak scaling: the frame gains 64*N samples per step). Deep bounces keep few paths alive, so many sample ids share a pass to
         keep 256 CUs busy: 4 / 8 / 16 / 32 ids per pass gave 2610 / 2769 / 2861 / 2907 Mrays/s on the hall (profiles/r02_ab_experiments.txt); rounds 2-5
         were quoted at 32, round 6 moved to 64 (same box, 32 -> 64: hall +2.1 %, Example-class +10.7 %, scan +14 % samples/s; profiles/r06_ab_experiments.txt).
Rays   = closest-hit rays + executed shadow rays + light-BVH queries (SURVEY.md §8d counting rule), counted on the device.
N > 1  = the frame is cut into 32x32 tiles dealt to the ranks by a lattice - tile (x, y) -> rank (x + k y) % N - (weak data-parallel over pixels, no collective while
         rendering); each rank accumulates its own pixels and ONE RCCL collective at the end assembles the frame moments on rank 0: a gather of
         the ranks' own tiles behind the C ABI (lumc_frame_gather; --reduce cabi-reduce: the reduce of zero-padded full frames it replaced).
Prints ONE JSON line on rank 0, as the LAST line of stdout, at most 4 KB (`headline`): the contract's fields, the dominant kernel's roofline, the three
big kernels in brief, value_exact (the bit-exact flavour on the same scene), the CPU baseline, the secondaries' values. Everything else - prose, ceilings,
L2 figures, per-ray counts, the secondaries' own blocks - goes to profiles/bench_detail.json (written by the same run) and to stderr.

roofline: the block of the kernel that took most of the timed region, chosen over ALL kernels; the three big kernels are also
reported by name (roofline_trace, roofline_shadow, roofline_shade). For a ray kernel
  achieved_algorithmic = (nodes*112 + triangles*48 + rays*40) / kernel time   SURVEY §8d's formula, counted in the kernel. Most of these
                         bytes are served by LDS/L1/L2, so this is NOT an HBM rate and no fraction of the HBM peak is derived from it.
  traffic              = memory-side bytes per launch of that kernel from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
                         same command and workload (profiles/pmc_counters.json, written by tools/pmc_collect.py; the factor applied to
                         FETCH_SIZE for this access pattern is calibrated by tools/microbench/fetch_calib.hip and stored in the file)
  achieved, frac       = traffic / average launch time, and that over the 8 TB/s HBM peak: the memory-side figure ("memory_side" says what
                         serves it: the hall's tree and triangles fit the 256 MiB Infinity Cache and FETCH_SIZE counts its hits, so there it
                         is a fabric rate; the 10 M-triangle scan is an HBM rate). null - never an inferred bound - unless the committed
                         counters were collected from this very source tree (source_hash) and workload.
  frac_of_dependent_gather_ceiling = memory-side line rate of the kernel over what tools/microbench/gather.hip reaches with chains of dependent
                         divergent 7 x 16-byte line gathers at the same occupancy and table size (profiles/gather_ceiling.json)
  l2                   = L2 request bytes per launch (TCP_TCC_READ_REQ x 64 B... see the file) over the aggregate L2 bandwidth
and bound "valu" for k_shade: wave-level VALU instructions per launch / time against the chip's VALU issue peak (its memory-side bytes are given too).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
L2_PEAK_GBPS = 34500.0   # MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate
# wave64 VALU instructions per second. The guide's vector peak (157.3 TFLOP/s f32) is one wave64 FMA per 2 cycles and SIMD: 256 CUs x 4 SIMDs x 2.4 GHz / 2
# = 1228.8 G/s; k_shade's `frac` is priced against that. tools/microbench/valu_rate.hip (profiles/r03_valu_rate.txt, 4 waves per SIMD) measures what single
# instruction kinds reach: v_fma_f32 (VOP3) 618-632 G/s, v_cmp+v_cndmask 644, two-operand VOP2 forms (v_mul_f32, v_fmac_f32) 846-924, v_pk_fma_f32 449-462 (two results
# each), v_rcp_f32 / v_sqrt_f32 300: one instruction's rate is not a peak, so the 614.4 figure is only quoted beside it (frac_of_measured_fma_rate).
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0          # MI355X_MICROARCH.md (157.3 TF vector f32 = 2 flops x 64 lanes x this): 1228.8 G wave64 instructions/s - the peak `frac` is priced against
VALU_MEASURED_FMA_GINST = 256 * 4 * 2.4 / 4.0  # what tools/microbench/valu_rate.hip measures for v_fma_f32 (VOP3): 614.4 G/s - reported as frac_of_measured_fma_rate, not as a peak
NODE_BYTES, TRI_BYTES = 112, 48  # one node visit fetches 7 x 16 B of a 128-B BVH4 node; triangle = 48 B (DESIGN.md "Algorithmic bytes")
IO_TRACE_BYTES = 24 + 4 + 12  # origin+dir, tmax, hit
IO_SHADOW_BYTES = 24 + 4 + 12  # origin+dir, tmax, RGB visibility
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_counters.json")

from luminary_amd.distributed import assemble_frame, tile_lattice_step, tile_pixels  # noqa: E402  (numpy only at import time)

WORKLOADS = {
    "example": "C2 Example-class scene (~100k triangles, 72 instances, 16 emissive quads)",
    "hall": "C3 Sponza-class hall (1.43M triangles, one mesh, 32 emissive panels)",
    "scan": "C5 scanned-object class (10.03M-triangle displaced icosphere, ground, 8 area lights)",
    "scan5m": "scanned-object class at 5.2M triangles (round-1 size)",
    "cornell": "C1 Cornell box (36 triangles)",
}


def build_workload(name, width, height, bounces):
    from luminary_amd import scenes
    if name == "example":
        return scenes.example_scene(width, height, bounces)
    if name == "hall":
        return scenes.hall_scene(width, height, bounces)
    if name == "scan":
        return scenes.scan_scene(width, height, bounces, triangles=10_000_000)
    if name == "scan5m":
        return scenes.scan_scene(width, height, bounces)
    if name == "cornell":
        return scenes.cornell_host("/tmp/lum_bench_cornell", width, height, bounces)
    raise SystemExit("unknown workload " + name)


def source_hash():
    """Identity of the device code a counter file belongs to: SHA-1 over the kernel sources, the core that launches them, the BVH builders and
    the build flags. tools/pmc_collect.py stores it with the counters; a bench run of another source tree reports no counter-based figure."""
    import hashlib
    from luminary_amd import build as b
    h = hashlib.sha1()
    csrc = os.path.join(ROOT, "luminary_amd", "csrc")
    files = sorted(os.path.join("device", f) for f in os.listdir(os.path.join(csrc, "device"))) + ["host/core.hip", "host/bvh_build.cpp", "host/lbvh.hip"]
    for f in files:
        with open(os.path.join(csrc, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    h.update(b._flags_identity().encode())
    return h.hexdigest()[:16]


def gather_ceiling(table_bytes, active_lanes=32):
    """Line rate (128-byte lines per second) tools/microbench/gather.hip measured for dependent divergent 7 x 16-byte gathers at 16 waves per CU with
    `active_lanes` of 64 lanes holding a ray, from the table size nearest to `table_bytes` (profiles/gather_ceiling.json); None without the file."""
    try:
        with open(os.path.join(ROOT, "profiles", "gather_ceiling.json")) as f:
            rows = [r for r in json.load(f)["rows"] if r["loads"] == 7 and r["group"] == 1 and r["dep"] == 1 and r["active_per_wave"] == active_lanes and r["waves_per_cu"] == 16]
        if not rows:
            return None
        r = min(rows, key=lambda r: abs(np.log(max(r["table_mib"] * 1048576.0, 1.0) / max(table_bytes, 1.0))))
        return {"lines_per_s": r["gvisits_per_s"] * 1e9, "table_mib": r["table_mib"], "line_tb_s": r["line_tb_s"]}
    except (OSError, KeyError, ValueError):
        return None


def pmc_record(workload, spp_per_step, flavour, lds_stack_bytes):
    """Per-kernel counter record of this workload from the committed PMC passes (None when there is none for this configuration).
    `_stale`: the passes ran another source tree (or another LDS split of the ray kernels): no counter-based figure is derived from them."""
    try:
        with open(PMC_FILE) as f:
            t = json.load(f)
        e = t["workloads"][workload]
        if int(e["spp_per_step"]) != int(spp_per_step) or e.get("flavour", "exact") != flavour:
            return None
        e = dict(e)
        e["_stale"] = int(e.get("lds_stack_bytes", 0)) != int(lds_stack_bytes) or e.get("source_hash") != source_hash()
        e["_source"] = "profiles/pmc_counters.json (%s)" % t.get("collected", "rocprofv3 --pmc passes of this command")
        e["_fetch_factor"] = t.get("fetch_size_factor")
        return e
    except (OSError, KeyError, ValueError):
        return None


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
            "sample_short": "oracle/ (scalar C restatement, OpenMP) on %d pixels x %d spp of the same frame, 9 depth passes: %.3g rays in %.1f s" % (npx, spp, rays, dt),
            "note": "unoptimised test oracle (scalar C restatement written for bit-exact checking, not for speed): a reported baseline, not a target",
            "sample": "oracle/ (CPU restatement, OpenMP over pixels, %d threads) on %d pixels x %d spp of the same frame, all 9 depth passes: "
                      "%.0f rays in %.1f s (its %.1f s BVH build excluded)" % (cores, npx, spp, rays, dt, t_build)}


def spawn_distributed(n):
    """`bench.py --gpus N` outside a torchrun environment: run the N-rank job as a child process and return its exit code. Nothing in
    this process has touched the GPU yet (no torch.cuda call, no HIP library loaded)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def run_workload(core, name, args, rank, world, dist, steps, warmup, want_output_chain, exact_steps=0):
    """Uploads workload `name`, times `steps` passes after `warmup` untimed ones (barrier + synchronize on both sides, max over ranks) and
    returns (json dict, device scene view)."""
    import torch
    from luminary_amd.core import CNT_AMBIENT_DEFERRED, CNT_AMBIENT_FALLBACK, CNT_LIGHT_BVH, CNT_NODES, CNT_SHADOW, CNT_TRACE, CNT_TRIS, CNT_VERTICES
    t_build = time.time()
    host = build_workload(name, args.width, args.height, args.bounces)
    label = WORKLOADS[name]
    if args.sky == "procedural":
        sky = host.get_sky()
        sky.mode = 0
        host.set_sky(sky)
        label += " under the procedural sky (mode DEFAULT)"
    if args.fog > 0.0:  # not a BASELINE configuration: the same scene inside the reference's fog volume (f4), for the kernel table
        fog = host.get_fog()
        fog.active, fog.density = True, args.fog
        host.set_fog(fog)
        label += " in fog of density %g" % args.fog
    if args.ocean is not None:  # not a BASELINE configuration either: the same scene with the reference's ocean at this height (f4)
        oc = host.get_ocean()
        oc.active, oc.height = True, args.ocean
        host.set_ocean(oc)
        label += " with the ocean at height %g" % args.ocean
    if args.clouds:  # likewise: the three cloud layers (procedural sky mode: the reference marches clouds there only)
        from luminary_amd import SKY_MODE_DEFAULT
        sky = host.get_sky()
        sky.mode = SKY_MODE_DEFAULT
        host.set_sky(sky)
        cl = host.get_cloud()
        cl.active = True
        host.set_cloud(cl)
        label += " under the procedural sky with clouds"
    view = host.device_scene()
    build_s = time.time() - t_build
    t_up = time.time()
    core.upload(view)  # builds the BVHs, generates the BSDF tables on the GPU
    upload_s = time.time() - t_up
    pixels = tile_pixels(view.width, view.height, rank, world) if dist is not None else None
    if pixels is None and args.pixel_tile:  # experiment: the frame's pixels dealt to the paths in t x t tiles instead of rows (the moments then come back in that order)
        import numpy as np
        t = args.pixel_tile
        ys, xs = np.divmod(np.arange(view.width * view.height, dtype=np.int64), view.width)
        pixels = np.lexsort((xs, ys, xs // t, ys // t)).astype(np.uint32)
    core.set_pixels(pixels)
    P = core.num_pixels
    frame_pixels = view.width * view.height
    stream = torch.cuda.current_stream().cuda_stream
    spp_step = args.samples_per_pass * world  # per-GPU paths per step stay W*H*samples_per_pass
    cabi = dist is not None and getattr(core, "_bench_comm", False)  # the library's own communicator is up (main): reduce behind the C ABI
    if cabi:
        fm = sm = None  # the context's own accumulators; lumc_frame_assemble scatters and reduces them

        def step(i):
            core.render(i * spp_step, spp_step, spp_step, 0, 0, stream)

        def assemble():
            if args.reduce == "cabi-reduce":
                core.frame_assemble(frame_pixels, 0, stream)   # every rank's zero-padded full frame, ncclReduce(SUM)
            else:
                core.frame_gather(view.width, view.height, 0, stream)  # the ranks' own pixels, one ncclGather (1 / N of the bytes)
    else:
        fm = torch.zeros(3 * P, dtype=torch.float32, device="cuda")
        sm = torch.zeros(P, dtype=torch.float32, device="cuda")

        def step(i):
            core.render(i * spp_step, spp_step, spp_step, fm.data_ptr(), sm.data_ptr(), stream)

        def assemble():
            assemble_frame(fm, sm, pixels, frame_pixels, dist, 0)

    for i in range(warmup):
        step(i)
    if dist is not None:
        assemble()  # untimed: first use of the communicator and its buffers
    torch.cuda.synchronize()
    core.synchronize()
    core.reset_counters()
    core.set_profiling(True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(steps):
        step(warmup + i)
    if dist is not None:
        # assemble the frame on rank 0: every rank scatters its pixels into a zero frame, one reduce over xGMI
        assemble()
    torch.cuda.synchronize()
    core.synchronize()
    local_elapsed = time.time() - t0  # this rank's own share (root: plus the frame's arrival), before it waits for the others
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.time() - t0

    cnt = core.counters()
    times = core.kernel_times()
    core.set_profiling(False)
    # output chain (tone map + ARGB8 of the frame just rendered), outside the timed region: streaming kernels, 40 B/pixel algorithmic
    output_chain = None
    if want_output_chain and dist is None:
        from luminary_amd.core import default_output_params
        core.set_profiling(True)
        op = default_output_params(view.width, view.height, max(spp_step * (steps + warmup), 1))
        for _ in range(5):
            core.generate_output(op, fm.data_ptr())
        out_ms, out_n = core.kernel_times()["output"]
        core.set_profiling(False)
        if out_n:
            per = out_ms / out_n
            output_chain = {"ms_per_frame": round(per, 4), "GB/s": round(40.0 * view.width * view.height / (per * 1e-3) / 1e9, 1),
                            "frac_of_hbm_peak": round(40.0 * view.width * view.height / (per * 1e-3) / 1e9 / HBM_PEAK_GBPS, 3)}
    # the bit-exact flavour (what every parity test runs) on the same uploaded scene: a driver-timed number for the path that is identical to the oracle
    exact = None
    if exact_steps > 0 and dist is None and core.flavour != "exact":
        was = core.flavour
        core.set_flavour("exact")
        step(0)
        torch.cuda.synchronize()
        core.synchronize()
        core.reset_counters()
        te = time.time()
        for i in range(exact_steps):
            step(1 + i)
        torch.cuda.synchronize()
        core.synchronize()
        te = time.time() - te
        ce = core.counters()
        exact = {"value": (ce[CNT_TRACE] + ce[CNT_SHADOW] + ce[CNT_LIGHT_BVH]) / te / 1e6, "unit": "Mrays/s", "flavour": "exact", "steps": exact_steps, "warmup": 1,
                 "ms_per_step": te / exact_steps * 1e3, "samples_per_s": view.width * view.height * spp_step * exact_steps / te}
        core.set_flavour(was)
    rays_local = cnt[CNT_TRACE] + cnt[CNT_SHADOW] + cnt[CNT_LIGHT_BVH]
    stats = torch.tensor([float(rays_local), float(cnt[CNT_TRACE]), float(cnt[CNT_SHADOW]), float(cnt[CNT_LIGHT_BVH]), elapsed,
                          float(cnt[CNT_AMBIENT_DEFERRED] - cnt[CNT_AMBIENT_FALLBACK])], dtype=torch.float64, device="cuda")
    if dist is not None:
        mx = stats.clone()
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        elapsed = float(mx[4])
    load_balance = None
    if dist is not None:  # what the first scaling record needs to explain itself: the deal's balance in rays and in time (efficiency <= mean / max)
        mine = torch.tensor([float(rays_local), local_elapsed * 1e3 / steps, float(P)], dtype=torch.float64, device="cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per = torch.stack(every).cpu().numpy()
        load_balance = {"rays": {"min": float(per[:, 0].min()), "mean": float(per[:, 0].mean()), "max": float(per[:, 0].max())},
                        "ms_per_step": {"min": float(per[:, 1].min()), "mean": float(per[:, 1].mean()), "max": float(per[:, 1].max())},
                        "pixels": {"min": int(per[:, 2].min()), "max": int(per[:, 2].max())},
                        "mean_over_max_rays": float(per[:, 0].mean() / max(per[:, 0].max(), 1.0)), "mean_over_max_ms": float(per[:, 1].mean() / max(per[:, 1].max(), 1e-9)),
                        "note": "per rank, before the closing barrier; rank 0's time includes the frame's arrival"}
    rays_total = float(stats[0])
    answered = float(stats[5])  # ambient samples answered by the next closest-hit ray: visibility queries that cost no traversal (not in `value`)
    del fm, sm

    # ---- rooflines (rank 0's kernels) ----
    pmc = pmc_record(name, args.samples_per_pass, core.flavour, core.lds_stack_bytes()) if (world == 1 and dist is None) else None
    nodes_trace, tris_trace, nodes_shadow, tris_shadow = cnt[CNT_NODES], cnt[CNT_TRIS], cnt[6], cnt[7]
    alg = {"trace": nodes_trace * NODE_BYTES + tris_trace * TRI_BYTES + cnt[CNT_TRACE] * IO_TRACE_BYTES,
           "shadow": nodes_shadow * NODE_BYTES + tris_shadow * TRI_BYTES + cnt[CNT_SHADOW] * IO_SHADOW_BYTES}
    pmc_name = {"trace": "k_trace", "shadow": "k_shadow_rays", "shade": "k_shade"}

    bvh = core.bvh_stats()  # BLAS nodes, BLAS triangles, TLAS nodes, light nodes
    working_set = (bvh[0] + bvh[2]) * 128.0 + bvh[1] * 48.0
    memory_side = ("fabric: the tree and triangles (%.0f MB) fit the 256 MiB Infinity Cache, whose hits FETCH_SIZE counts" if working_set < 240e6 else
                   "hbm: the tree and triangles (%.0f MB) exceed the 256 MiB Infinity Cache") % (working_set / 1e6)

    def add_traffic(r, e, avg_ms):
        """memory-side bytes per launch from the counter record `e` of the same source tree; stale records only say that they are stale"""
        if not e or avg_ms <= 0:
            return
        if pmc["_stale"]:
            r["traffic_stale"] = {"note": "profiles/pmc_counters.json was collected from another source tree or LDS split: no counter-based figure for this build",
                                  "source_hash_then": pmc.get("source_hash"), "source_hash_now": source_hash()}
            return
        r["traffic"] = e["bytes_per_launch"]
        r["traffic_fetch"], r["traffic_write"] = e["fetch_bytes_per_launch"], e["write_bytes_per_launch"]
        r["memory_side"] = memory_side
        r["traffic_source"], r["fetch_size_factor"], r["source_hash"] = pmc["_source"], pmc["_fetch_factor"], pmc.get("source_hash")
        if e.get("l2_read_bytes_per_launch"):
            l2 = e["l2_read_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
            r["l2"] = {"achieved": l2, "peak": L2_PEAK_GBPS, "frac": l2 / L2_PEAK_GBPS, "hit_rate": e.get("l2_hit_rate"), "unit": "GB/s"}

    def ray_roofline(k):
        ms, n = times[k]
        avg_ms = ms / max(n, 1)
        r = {"bound": "hbm", "kernel": pmc_name[k], "launches": n, "avg_launch_ms": avg_ms, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
             "achieved_algorithmic": alg[k] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0, "algorithmic_bytes_per_launch": alg[k] / max(n, 1),
             "achieved": None, "frac": None, "traffic": None, "l2": None}
        e = pmc.get(pmc_name[k]) if pmc else None
        add_traffic(r, e, avg_ms)
        if r["traffic"] is not None:
            # the PMC passes ran the same passes (same sample ids) as this run's steps: bytes per launch carry over, time is this run's
            r["achieved"] = r["traffic"] / (avg_ms * 1e-3) / 1e9
            r["frac"] = r["achieved"] / HBM_PEAK_GBPS
            r["lane_utilisation"], r["wait_fraction"] = e.get("valu_lane_utilisation"), e.get("wait_fraction")
            ceil = gather_ceiling(working_set)
            if ceil:
                lines_per_s = r["traffic_fetch"] / 128.0 / (avg_ms * 1e-3)
                r["frac_of_dependent_gather_ceiling"] = lines_per_s / ceil["lines_per_s"]
                r["dependent_gather_ceiling"] = dict(ceil, note="tools/microbench/gather.hip: dependent 7 x 16 B gathers of random 128-B lines, 16 waves per CU, "
                                                     "32 of 64 lanes active, every visit a miss; the kernel's figure counts only the lines that reach the memory side")
        return r

    def valu_roofline(k):
        ms, n = times[k]
        avg_ms = ms / max(n, 1)
        r = {"bound": "valu", "kernel": pmc_name[k], "launches": n, "avg_launch_ms": avg_ms, "peak": VALU_PEAK_GINST, "unit": "G wave64 VALU instructions/s",
             "achieved": None, "frac": None, "traffic": None, "vertices_per_launch": cnt[CNT_VERTICES] / max(n, 1)}
        e = pmc.get(pmc_name[k]) if pmc else None
        add_traffic(r, e, avg_ms)
        if r["traffic"] is not None and e.get("valu_insts_per_launch"):
            r["achieved"] = e["valu_insts_per_launch"] / (avg_ms * 1e-3) / 1e9
            r["frac"] = r["achieved"] / VALU_PEAK_GINST
            r["frac_of_measured_fma_rate"] = r["achieved"] / VALU_MEASURED_FMA_GINST
            r["valu_insts_per_launch"] = e["valu_insts_per_launch"]
            r["lane_utilisation"], r["wait_fraction"] = e.get("valu_lane_utilisation"), e.get("wait_fraction")
            r["valu_insts_per_vertex_lane"] = e["valu_insts_per_launch"] * 64.0 * (e.get("valu_lane_utilisation") or 0.0) / max(r["vertices_per_launch"], 1.0)
            r["memory_side_GBps"] = r["traffic"] / (avg_ms * 1e-3) / 1e9
        return r

    blocks = {"trace": ray_roofline("trace"), "shadow": ray_roofline("shadow"), "shade": valu_roofline("shade")}
    dominant = max(blocks, key=lambda k: times[k][0])  # over ALL kernels: nothing outside these three comes close (kernel_share_rank0)
    roofline = dict(blocks[dominant], dominant_of="all kernels of the timed region by total time")
    total_ms = sum(v[0] for v in times.values()) or 1.0
    out = {
        "value": rays_total / elapsed / 1e6, "unit": "Mrays/s", "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "config": {"workload": label, "width": view.width, "height": view.height, "max_ray_depth": view.max_ray_depth, "flavour": core.flavour, "lds_stack_bytes": core.lds_stack_bytes(), "source_hash": source_hash(), "ray_sorting": core.ray_sorting,
                   "spp_per_step": spp_step, "paths_per_gpu_per_step": P * spp_step, "partition": "32x32 image tiles, tile (x, y) -> rank (x + %d y) %% %d (lumc_tile_owner)" % (tile_lattice_step(world), world) if dist is not None else "single GPU",
                   "load_balance": load_balance,
                   "frame_reduce": None if dist is None else (("C ABI: lumc_frame_assemble (RCCL ncclReduce of full frames)" if args.reduce == "cabi-reduce" else
                                                               "C ABI: lumc_frame_gather (RCCL ncclGather of the ranks' own tiles)") if cabi else "torch.distributed.reduce (RCCL)"),
                   "rccl_ranks": None if dist is None else (core.comm_count() if cabi else dist.get_world_size()),  # ncclCommCount of the library's own communicator
                   "samples_per_s": view.width * view.height * spp_step * steps / elapsed,
                   "seconds_to_1024spp": 1024.0 / (spp_step * steps / elapsed),
                   "rays": {"closest": float(stats[1]), "shadow": float(stats[2]), "light_bvh": float(stats[3])},
                   # `value` counts executed rays only (SURVEY 8d's rule). Ambient samples whose visibility the path's next closest-hit ray answered cost no
                   # traversal and are not in it; the reference would have traced each of them: value + these = the rate in the reference's own ray count.
                   "ambient_reuse": core.ambient_reuse, "rays_answered_without_trace": answered, "mrays_per_s_answered": (rays_total + answered) / elapsed / 1e6,
                   "value_note": ("executed rays only; %.3g ambient visibility queries were answered by the paths' next closest-hit rays and are not in it "
                                  "(rounds 1-3 traced them): in the reference's ray count the rate is mrays_per_s_answered" % answered) if answered > 0 else None,
                   "per_ray_rank0": {"nodes_closest": round(nodes_trace / max(cnt[CNT_TRACE], 1), 2), "tris_closest": round(tris_trace / max(cnt[CNT_TRACE], 1), 2),
                                     "nodes_shadow": round(nodes_shadow / max(cnt[CNT_SHADOW], 1), 2), "tris_shadow": round(tris_shadow / max(cnt[CNT_SHADOW], 1), 2),
                                     "lds_hit_rate_closest": round(cnt[10] / max(nodes_trace, 1), 3), "lds_hit_rate_shadow": round(cnt[11] / max(nodes_shadow, 1), 3)},
                   "kernel_ms_rank0": {k: round(v[0], 3) for k, v in times.items()},
                   "kernel_share_rank0": {k: round(v[0] / total_ms, 3) for k, v in times.items() if v[0] > 0},
                   "output_chain_rank0": output_chain, "scene_build_s": round(build_s, 2), "scene_upload_s": round(upload_s, 2)},
        "roofline": roofline, "roofline_trace": blocks["trace"], "roofline_shadow": blocks["shadow"], "roofline_shade": blocks["shade"],
    }
    out["value_exact"] = exact
    return out, view


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="hall", choices=sorted(WORKLOADS))
    ap.add_argument("--secondary", default="example,scan", help="workloads timed after the headline at N = 1 and reported under 'secondary' ('none' skips them)")
    ap.add_argument("--secondary-steps", type=int, default=4)
    ap.add_argument("--exact-steps", type=int, default=4, help="N = 1: steps of the headline workload timed once more in the bit-exact flavour (value_exact; 0 or --secondary none skip it)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--samples-per-pass", type=int, default=64, help="sample ids per wavefront pass = per step")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU baseline work (0 = skip)")
    ap.add_argument("--flavour", default=None, choices=["fast", "exact"], help="arithmetic flavour of the device code (default: the library's, fast)")
    ap.add_argument("--reduce", default="cabi", choices=["cabi", "cabi-reduce", "torch"],
                    help="N > 1: who assembles the frame on rank 0 - the library behind the C ABI with one ncclGather of the ranks' own tiles (lumc_frame_gather, default) or "
                         "with an ncclReduce of zero-padded full frames (lumc_frame_assemble), or torch.distributed's reduce")
    ap.add_argument("--ambient-reuse", default="auto", choices=["auto", "on", "off"],
                    help="ambient samples answered by the next closest-hit ray instead of a visibility ray (lumc_set_ambient_reuse; auto = the flavour's default: fast on, exact off)")
    ap.add_argument("--sort", type=int, default=None, choices=[0, 1, 2, 3], help="ray ordering between bounces: 0 queue order, 1 closest-hit rays sorted, 2 visibility rays too, 3 path queue physically reordered")
    ap.add_argument("--pixel-tile", type=int, default=0, help="experiment: order the paths by t x t pixel tiles instead of pixel rows (N = 1)")
    ap.add_argument("--clouds", action="store_true", help="procedural sky with the three cloud layers active (not a BASELINE configuration)")
    ap.add_argument("--ocean", type=float, default=None, help="height of an ocean surface put into the scene (default: none, the BASELINE configurations)")
    ap.add_argument("--fog", type=float, default=0.0, help="density of the fog volume the scene is put in (0 = none, the BASELINE configurations)")
    ap.add_argument("--sky", default="constant", choices=["constant", "procedural"],
                    help="constant = the benchmark settings (SURVEY §8d); procedural = sky mode DEFAULT: ray-marched atmosphere and sun sampling")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_distributed(args.gpus))
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "WORLD_SIZE" in os.environ:  # under torchrun the distributed path runs even with one rank (exercises it on a single-GPU box)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from luminary_amd.core import Core
    core = Core(local_rank)
    if args.flavour:
        core.set_flavour(args.flavour)
    if args.sort is not None:
        core.set_ray_sorting(args.sort)
    core.set_ambient_reuse({"auto": -1, "on": 1, "off": 0}[args.ambient_reuse])
    if dist is not None and args.reduce != "torch":
        # the library's own RCCL communicator: rank 0 makes the id, torch.distributed only carries its 128 bytes to the other ranks
        try:
            ids = [Core.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            core.comm_init_rank(world, rank, ids[0])
            ok = torch.ones(1, device="cuda")
        except Exception as e:  # noqa: BLE001 - reported by every rank, then the run ends non-zero: `--reduce torch` is the way to ask for torch's reduce
            sys.stderr.write("rank %d: C-ABI communicator failed (%s)\n" % (rank, e))
            ok = torch.zeros(1, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        core._bench_comm = bool(ok.item() > 0)
        if not core._bench_comm:
            dist.destroy_process_group()
            raise SystemExit("bench.py: the library's RCCL communicator (lumc_comm_init_rank) could not be created on every rank; "
                             "run with --reduce torch to reduce through torch.distributed instead")
        if core.comm_count() != world:
            dist.destroy_process_group()
            raise SystemExit("bench.py: the library's communicator reports %d ranks, the job has %d" % (core.comm_count(), world))
    # value_exact belongs to the full report, like the secondaries: `--secondary none` (profiling passes, A/B tools) times the headline flavour alone
    head, view = run_workload(core, args.workload, args, rank, world, dist, args.steps, args.warmup, True, exact_steps=args.exact_steps if args.secondary != "none" else 0)
    exact = head.pop("value_exact", None)
    # reported at N=1 only; the oracle needs the sky tables for the procedural sky, which the bench does not generate on the CPU
    cpu = cpu_baseline(view, args.cpu_budget) if (args.cpu_budget > 0 and dist is None and args.sky == "constant" and rank == 0) else None
    secondary = {}
    if dist is None and args.secondary != "none":
        for name in [s for s in args.secondary.split(",") if s and s != args.workload]:
            sec, _ = run_workload(core, name, args, rank, world, None, args.secondary_steps, 1, False)
            sec.pop("value_exact", None)
            secondary[name] = sec
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    try:  # how far the benchmarked flavour is from the bit-exact one on this scene (tests/test_flavours.py gates it on the GPU)
        with open(os.path.join(ROOT, "profiles", "flavour_gate.json")) as f:
            gate = json.load(f)
        if head["config"]["flavour"] == "fast" and args.workload == "hall":
            head["config"]["fast_vs_exact_rel_l2_1024spp"] = gate["fast_vs_exact_rel_l2"]["1024"]
            head["config"]["fast_vs_exact_note"] = gate["note"]
            conv = gate.get("against_converged_render")
            if conv:  # the like-for-like check against an exact render of 16384 spp (tools/flavour_gate.py, tests/test_flavours.py)
                head["config"]["flavour_gate"] = {k: conv[k] for k in ("truth_spp", "spp", "e_fast", "e_exact", "e_independent_exact", "e_fast_over_e_exact")}
    except (OSError, KeyError, ValueError):
        pass
    detail = {"metric": METRIC, "value": head["value"], "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
              "config": head["config"], "roofline": head["roofline"], "roofline_trace": head["roofline_trace"], "roofline_shadow": head["roofline_shadow"],
              "roofline_shade": head["roofline_shade"], "value_exact": exact, "cpu_baseline": cpu, "secondary": secondary or None}
    # Everything goes to the sidecar (and to stderr); stdout's LAST line is the headline alone, a few KB: the driver keeps the tail of stdout only.
    detail_path = os.path.join(ROOT, "profiles", "bench_detail.json")
    try:
        with open(detail_path, "w") as f:
            json.dump(detail, f, indent=1)
            f.write("\n")
    except OSError as e:
        sys.stderr.write("bench.py: could not write %s (%s)\n" % (detail_path, e))
    sys.stderr.write("bench detail: " + json.dumps(detail) + "\n")
    sys.stderr.flush()
    line = json.dumps(headline(detail))
    assert len(line) <= 4096, "headline grew to %d bytes" % len(line)
    print(line)
    sys.stdout.flush()


METRIC = "Mrays/s at 1920x1080, 8 bounces"


def _r(x, digits=5):
    """numbers of the headline carry `digits` significant digits"""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    return x


def headline(d):
    """The one JSON line the driver parses: the contract's fields, the dominant kernel's roofline in full, the three big kernels in brief, the CPU
    baseline, the secondaries' values. Prose, ceilings, L2 figures, per-ray counts and the secondaries' blocks live in profiles/bench_detail.json."""
    c = d["config"]
    cfg_keys = ["workload", "width", "height", "max_ray_depth", "flavour", "lds_stack_bytes", "spp_per_step", "paths_per_gpu_per_step", "partition", "frame_reduce", "rccl_ranks",
                "samples_per_s", "seconds_to_1024spp", "rays", "rays_answered_without_trace", "mrays_per_s_answered", "ambient_reuse", "value_note", "bvh", "source_hash",
                "fast_vs_exact_rel_l2_1024spp", "kernel_share_rank0"]
    cfg = {k: c[k] for k in cfg_keys if c.get(k) is not None}

    def brief(r, full=False):
        keys = ["kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches"] if full else ["kernel", "bound", "frac", "traffic", "avg_launch_ms"]
        out = {k: r.get(k) for k in keys}
        for k in ("frac_of_measured_fma_rate", "frac_of_dependent_gather_ceiling", "lane_utilisation", "wait_fraction"):
            if r.get(k) is not None:
                out[k] = r[k]
        if r.get("traffic") is None and r.get("traffic_stale"):
            out["traffic_stale"] = "counters in profiles/pmc_counters.json are of source %s, this build is %s" % (r["traffic_stale"].get("source_hash_then"), r["traffic_stale"].get("source_hash_now"))
        return out

    h = {k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    h["config"] = cfg
    h["roofline"] = brief(d["roofline"], True)
    for k in ("roofline_trace", "roofline_shadow", "roofline_shade"):
        h[k] = brief(d[k])
    if d.get("value_exact"):
        h["value_exact"] = d["value_exact"]
    if d.get("cpu_baseline"):
        b = d["cpu_baseline"]
        h["cpu_baseline"] = {"value": b["value"], "unit": b["unit"], "cores": b["cores"], "kind": b["kind"], "cpu": b.get("cpu"), "sample": b.get("sample_short", b.get("sample", ""))[:160]}
    if d.get("secondary"):
        h["secondary"] = {n: {"value": s["value"], "ms_per_step": s["ms_per_step"], "steps": s["steps"], "samples_per_s": s["config"].get("samples_per_s")} for n, s in d["secondary"].items()}
    h["detail"] = "profiles/bench_detail.json"

    def rnd(o):
        if isinstance(o, dict):
            return {k: rnd(v) for k, v in o.items()}
        if isinstance(o, list):
            return [rnd(v) for v in o]
        return _r(o)
    return rnd(h)


if __name__ == "__main__":
    main()
