#!/usr/bin/env python3
"""Collects the per-kernel hardware counters bench.py's roofline block quotes, on the GPU box (through gpurun):

  python tools/pmc_collect.py <out_dir> [--workloads hall,example,scan] [--flavour fast|exact] [--calib]

For every workload it runs `python3 bench.py --workload W --steps 2 --warmup 1 --cpu-budget 0 --secondary none` under rocprofv3 once per
counter group (separate --pmc passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit
one pass) and writes <out_dir>/pmc_counters.json, which is then committed as profiles/pmc_counters.json.
--calib additionally runs tools/microbench/fetch_calib.hip under the FETCH_SIZE / TCC_EA0_RDREQ / TCP passes and stores the factor that
turns FETCH_SIZE into bytes for divergent 16-byte gathers ("fetch_size_factor"); without it the previous file's factor is kept.

Units: FETCH_SIZE / WRITE_SIZE are reported in KiB. On gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B although requests can be 128 B
(coalesced streams: factor 2, the guide's figure). The factor for this kernel's pattern is whatever the calibration measures.
"""
import argparse
import csv
import glob
import json
import os
import re
import subprocess
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PASSES = [
    ("fetch", ["FETCH_SIZE"]),
    ("write", ["WRITE_SIZE"]),
    ("tcc", ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCP_TCC_READ_REQ_sum"]),
    ("ea", ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_sum"]),
    ("sq", ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES", "SQ_INSTS_LDS"]),
    # the address unit and the L1: few hardware counters per block instance, so small groups (six TA counters in one pass: "Request exceeds the
    # capabilities of the hardware to collect", and rocprofv3 then hangs in its signal handler - every pass runs under a time limit)
    ("ta", ["TA_BUSY_avr", "GRBM_GUI_ACTIVE"]),
    ("ta2", ["TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum"]),
    ("ta3", ["TA_FLAT_READ_WAVEFRONTS_sum", "TA_BUSY_max"]),
    ("tcp", ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_PENDING_STALL_CYCLES_sum"]),
    ("tcp2", ["TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum", "TCP_GATE_EN1_sum"]),
    ("sq2", ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_SMEM"]),
]


PASS_TIMEOUT_S = 420


def kernel_key(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"<.*", "", name)
    return name.split("::")[-1]


def full_key(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name).split("::")[-1]


def run_pass(out_dir, tag, counters, cmd, key_fn):
    d = os.path.join(out_dir, tag)
    os.makedirs(d, exist_ok=True)
    full = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", d, "--"] + cmd
    t = time.time()
    with open(os.path.join(out_dir, tag + ".log"), "w") as log:
        proc = subprocess.Popen(full, stdout=log, stderr=subprocess.STDOUT, cwd=ROOT, start_new_session=True)  # its own process group: the one thing killed on a time-out
        try:
            rc = proc.wait(timeout=PASS_TIMEOUT_S)
        except subprocess.TimeoutExpired:
            import signal
            os.killpg(proc.pid, signal.SIGKILL)
            proc.wait()
            rc = -9
    print("pass %-28s rc %d  %.0f s" % (tag, rc, time.time() - t), flush=True)
    files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    tot, disp = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    for f in files[-1:]:
        for row in csv.DictReader(open(f)):
            k = key_fn(row["Kernel_Name"])
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[k].add(row["Dispatch_Id"])
    return {k: dict(v, _launches=len(disp[k])) for k, v in tot.items()}  # counter totals over all launches + the launch count


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir")
    ap.add_argument("--workloads", default="hall,example,scan")
    ap.add_argument("--flavour", default=None)
    ap.add_argument("--calib", action="store_true")
    ap.add_argument("--spp", type=int, default=64)  # bench.py's default
    ap.add_argument("--extra", default="", help="extra bench.py arguments for every workload (e.g. '--clouds'): feature frames instead of the BASELINE ones")
    ap.add_argument("--passes", default="", help="comma-separated subset of the counter groups (fetch,write,tcc,ea,sq,ta,tcp,sq2); default: all")
    ap.add_argument("--label", default="", help="suffix of the workload keys when --extra is given (e.g. 'clouds' -> 'example+clouds')")
    args = ap.parse_args()
    os.makedirs(args.out_dir, exist_ok=True)
    os.environ.setdefault("TMPDIR", "/tmp")
    prev = {}
    try:
        prev = json.load(open(os.path.join(ROOT, "profiles", "pmc_counters.json")))
    except (OSError, ValueError):
        pass
    out = {"collected": "rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --workload W --steps 2 --warmup 1 --cpu-budget 0 --secondary none; "
                        "one pass per counter group; tools/pmc_collect.py", "workloads": {}, "fetch_size_factor": prev.get("fetch_size_factor"),
           "calibration": prev.get("calibration")}

    if args.calib:
        exe = os.path.join(args.out_dir, "fetch_calib")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, os.path.join(ROOT, "tools", "microbench", "fetch_calib.hip")])
        cal = {}
        for tag, counters in PASSES[:4]:
            res = run_pass(args.out_dir, "calib_" + tag, counters, [exe], full_key)
            for k, v in res.items():
                if "k_calib" in k:
                    cal.setdefault(k, {}).update({c: x for c, x in v.items()})
        lines_cold, threads_hot = float(1 << 25), 65536.0 * 256.0 * 64.0
        summary = {}
        for k, v in sorted(cal.items()):
            m = re.search(r"k_calib<(\d+), *(\d+)>", k)
            n16, hot = int(m.group(1)), int(m.group(2))
            e = {"counters": v}
            if not hot:
                fetch = v.get("FETCH_SIZE", 0.0) * 1024.0
                e["fetch_size_bytes"] = fetch
                e["line_bytes_touched"] = lines_cold * 128.0
                e["useful_bytes"] = lines_cold * 16.0 * (n16 if n16 else 8)
                e["factor_to_line_bytes"] = lines_cold * 128.0 / fetch if fetch else None
                e["rdreq_per_line"] = v.get("TCC_EA0_RDREQ_sum", 0.0) / lines_cold
                e["rdreq_32b_per_line"] = v.get("TCC_EA0_RDREQ_32B_sum", 0.0) / lines_cold
                e["tcp_tcc_read_req_per_line"] = v.get("TCP_TCC_READ_REQ_sum", 0.0) / lines_cold
                e["tcc_req_per_line"] = v.get("TCC_REQ_sum", 0.0) / lines_cold
            else:
                loads = threads_hot * n16
                e["tcp_tcc_read_req_per_lane_load"] = v.get("TCP_TCC_READ_REQ_sum", 0.0) / loads
                e["tcc_req_per_lane_load"] = v.get("TCC_REQ_sum", 0.0) / loads
                e["fetch_size_bytes"] = v.get("FETCH_SIZE", 0.0) * 1024.0
            summary[k] = e
        out["calibration"] = summary
        g7 = summary.get("k_calib<7, 0>") or summary.get("k_calib<7,0>")
        if g7 and g7.get("factor_to_line_bytes"):
            out["fetch_size_factor"] = {"value": g7["factor_to_line_bytes"], "pattern": "7 x 16 B of one random 128-B line per lane, every line once, 4 GiB table",
                                        "coalesced_stream": (summary.get("k_calib<0, 0>") or {}).get("factor_to_line_bytes")}
        json.dump(out, open(os.path.join(args.out_dir, "pmc_counters.json"), "w"), indent=1, sort_keys=True)

    factor = (out.get("fetch_size_factor") or {}).get("value") or 2.0
    for w in [x for x in args.workloads.split(",") if x]:
        cmd = ["python3", "bench.py", "--workload", w, "--steps", "2", "--warmup", "1", "--cpu-budget", "0", "--secondary", "none", "--samples-per-pass", str(args.spp)]
        if args.flavour:
            cmd += ["--flavour", args.flavour]
        if args.extra:
            cmd += args.extra.split()
        wkey = w + ("+" + args.label if args.label else "")
        merged = defaultdict(dict)  # kernel -> counter -> value PER LAUNCH (every pass launches the same kernels the same number of times)
        for tag, counters in PASSES:
            if args.passes and tag not in args.passes.split(","):
                continue
            res = run_pass(args.out_dir, "%s_%s" % (wkey, tag), counters, cmd, kernel_key)
            for k, v in res.items():
                if not k.startswith("k_"):
                    continue
                n = max(v.pop("_launches"), 1)
                merged[k]["_launches"] = n
                for c, x in v.items():
                    merged[k][c] = x / n
        flavour, lds_stack_bytes, source_hash = args.flavour, 0, None
        try:
            line = [l for l in open(os.path.join(args.out_dir, "%s_fetch.log" % wkey)) if l.startswith("{")][-1]
            config = json.loads(line)["config"]
            flavour = config.get("flavour", flavour)
            lds_stack_bytes = int(config.get("lds_stack_bytes", 0))
            source_hash = config.get("source_hash")
        except (OSError, IndexError, ValueError, KeyError):
            pass
        # the build the counters belong to: bench.py only prices the ray kernels with them when its library has the same LDS split
        rec = {"spp_per_step": args.spp, "flavour": flavour or "exact", "lds_stack_bytes": lds_stack_bytes, "source_hash": source_hash}
        for key, v in sorted(merged.items()):
            n = v.get("_launches", 1)
            fetch, write = v.get("FETCH_SIZE", 0.0) * 1024.0 * factor, v.get("WRITE_SIZE", 0.0) * 1024.0
            e = {"launches": n, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "bytes_per_launch": fetch + write,
                 "fetch_size_raw_bytes_per_launch": v.get("FETCH_SIZE", 0.0) * 1024.0}
            if v.get("TCC_HIT_sum") is not None and (v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)) > 0:
                e["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
                e["l2_requests_per_launch"] = v.get("TCC_REQ_sum", 0.0)
                e["l1_to_l2_read_requests_per_launch"] = v.get("TCP_TCC_READ_REQ_sum", 0.0)
                e["l2_read_bytes_per_launch"] = v.get("TCP_TCC_READ_REQ_sum", 0.0) * 64.0  # 64 B per L1->L2 read request (checked by the calibration's hot-table rows)
            if v.get("TCC_EA0_RDREQ_sum"):
                e["ea_rdreq_per_launch"] = v["TCC_EA0_RDREQ_sum"]
                e["ea_rdreq_32b_per_launch"] = v.get("TCC_EA0_RDREQ_32B_sum", 0.0)
                e["ea_rdreq_dram_per_launch"] = v.get("TCC_EA0_RDREQ_DRAM_sum", 0.0)
                e["ea_wrreq_per_launch"] = v.get("TCC_EA0_WRREQ_sum", 0.0)
            if v.get("SQ_INSTS_VALU"):
                e["valu_insts_per_launch"] = v["SQ_INSTS_VALU"]
                e["valu_lane_utilisation"] = v.get("SQ_THREAD_CYCLES_VALU", 0.0) / (v["SQ_ACTIVE_INST_VALU"] * 64.0) if v.get("SQ_ACTIVE_INST_VALU") else None
                e["waves_per_launch"] = v.get("SQ_WAVES", 0.0)
                e["wait_fraction"] = v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else None
                e["wave_cycles_per_launch"] = v.get("SQ_WAVE_CYCLES", 0.0) * 4.0  # quad-cycles -> cycles
                e["busy_cycles_per_launch"] = v.get("SQ_BUSY_CYCLES", 0.0)
                for c in ("SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
                    if v.get(c) is not None:
                        e[c.lower().replace("sq_", "") + "_per_launch"] = v[c]
            if v.get("GRBM_GUI_ACTIVE"):
                # the vector-memory address unit: one divergent 16-byte lane load per cycle and CU (tools/microbench/gather.hip), so its busy share says how
                # close a ray kernel is to the rate at which its lanes can ask for nodes and triangles at all
                e["ta_busy_frac_avg"], e["ta_busy_frac_max"] = v.get("TA_BUSY_avr", 0.0) / v["GRBM_GUI_ACTIVE"], v.get("TA_BUSY_max", 0.0) / v["GRBM_GUI_ACTIVE"]
                e["gui_active_cycles_per_launch"] = v["GRBM_GUI_ACTIVE"]
                e["ta_addr_stalled_by_tc_cycles_per_launch"] = v.get("TA_ADDR_STALLED_BY_TC_CYCLES_sum", 0.0)
                e["ta_data_stalled_by_tc_cycles_per_launch"] = v.get("TA_DATA_STALLED_BY_TC_CYCLES_sum", 0.0)
                e["ta_flat_read_wavefronts_per_launch"] = v.get("TA_FLAT_READ_WAVEFRONTS_sum", 0.0)
            if v.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
                e["l1_cache_accesses_per_launch"] = v["TCP_TOTAL_CACHE_ACCESSES_sum"]
                e["l1_ta_data_stall_cycles_per_launch"] = v.get("TCP_TCP_TA_DATA_STALL_CYCLES_sum", 0.0)
                e["l1_pending_stall_cycles_per_launch"] = v.get("TCP_PENDING_STALL_CYCLES_sum", 0.0)
                e["l1_read_tagconflict_stall_cycles_per_launch"] = v.get("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", 0.0)
                e["l1_busy_cycles_per_launch"] = v.get("TCP_GATE_EN1_sum", 0.0)
                if v.get("TCP_TCC_READ_REQ_sum"):
                    e["l1_to_l2_read_latency_cycles"] = v.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / v["TCP_TCC_READ_REQ_sum"]
            rec[key] = e
        if args.extra:
            rec["bench_args"] = args.extra
        out["workloads"][wkey] = rec
        json.dump(out, open(os.path.join(args.out_dir, "pmc_counters.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({w: {k: {"MB/launch": round(e["bytes_per_launch"] / 1e6, 1), "l2_hit": e.get("l2_hit_rate")} for k, e in r.items() if isinstance(e, dict)}
                      for w, r in out["workloads"].items()}, indent=1))


if __name__ == "__main__":
    main()
