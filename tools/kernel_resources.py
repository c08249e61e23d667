#!/usr/bin/env python3
"""Table of registers / scratch / LDS per kernel from the last build (luminary_amd/lib/obj/kernel_resource_usage.txt; --variant NAME: of that variant's build)."""
import os, re, subprocess, sys
p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "luminary_amd", "lib", "obj", "kernel_resource_usage.txt")
if len(sys.argv) > 2 and sys.argv[1] == "--variant":
    p = os.path.join(os.path.dirname(os.path.dirname(p)), "variants", sys.argv[2], "obj", "kernel_resource_usage.txt")
    del sys.argv[1:3]
rows, cur = [], None
for line in open(p):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    if cur is None: continue
    for key in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m: cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
print("%-46s %5s %5s %7s %4s %6s %6s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "sSpill", "vSpill", "LDS"))
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    if len(sys.argv) > 1 and not any(a in n for a in sys.argv[1:]): continue
    print("%-46s %5d %5d %7d %4d %6d %6d %6d" % (n[:46], r.get("VGPRs", 0), r.get("TotalSGPRs", 0), r.get("ScratchSize [bytes/lane]", 0), r.get("Occupancy [waves/SIMD]", 0),
                                                 r.get("SGPRs Spill", 0), r.get("VGPRs Spill", 0), r.get("LDS Size [bytes/block]", 0)))
