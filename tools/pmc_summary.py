#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv files per kernel: python tools/pmc_summary.py gpurun_out/<tag>_sq [more dirs]"""
import csv, glob, re, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
for d in sys.argv[1:]:
    import os
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]:  # newest pass only
        seen = set()
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("lum::", "").replace("void ", "")[:28]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (k, row["Dispatch_Id"])
            if key not in seen and row["Counter_Name"].startswith(("SQ_WAVES", "SQ_INSTS_VMEM_RD", "TCC_HIT")):
                seen.add(key); calls[k] += 1
names = sorted({c for k in acc for c in acc[k]})
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get("SQ_INSTS_VMEM_RD", 0))):
    a = acc[k]
    line = f"{k:28s}"
    for c in names:
        line += f" {c.replace('SQ_', '').replace('_sum', '')}={a.get(c, 0):.4g}"
    if a.get("SQ_ACTIVE_INST_VALU"):
        line += f" | lane_util={a['SQ_THREAD_CYCLES_VALU'] / (a['SQ_ACTIVE_INST_VALU'] * 64):.3f}"
    if a.get("SQ_WAVES"):
        line += f" valu/wave={a['SQ_INSTS_VALU'] / a['SQ_WAVES']:.0f}"
    if a.get("SQ_WAVE_CYCLES"):
        line += f" wait_frac={a.get('SQ_WAIT_ANY', 0) / a['SQ_WAVE_CYCLES']:.2f}"
    print(line)
