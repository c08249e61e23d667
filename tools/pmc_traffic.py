#!/usr/bin/env python3
"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py` into profiles/pmc_traffic.json (read by bench.py for
roofline.traffic).

  python tools/pmc_traffic.py <workload> <spp_per_step> <fetch_dir> <write_dir>

FETCH_SIZE/WRITE_SIZE are reported in kilobytes; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled
(MI355X_MICROARCH.md, "HBM"). Memory-side requests served by the Infinity Cache are included, so this is an upper bound of HBM bytes.
"""
import csv, glob, json, os, re, sys
from collections import defaultdict

workload, spp, fetch_dir, write_dir = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]


def per_kernel(d, counter):
    tot, n = defaultdict(float), defaultdict(set)
    files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for f in files[-1:]:  # gpurun merges successive runs into the same directory: only the newest pass counts
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("lum::", "").replace("void ", "")
            tot[k] += float(row["Counter_Value"])
            n[k].add(row["Dispatch_Id"])
    return {k: (tot[k], len(n[k])) for k in tot}


fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
try:
    out = json.load(open(out_path))
except OSError:
    out = {}
out[workload] = {}
for k, (kb, launches) in sorted(fetch.items()):
    if not k.startswith("k_"):
        continue
    wkb = write.get(k, (0.0, launches))[0]
    out[workload][k] = {"spp_per_step": spp, "launches": launches, "fetch_bytes_per_launch": 2.0 * kb * 1024.0 / launches,
                        "write_bytes_per_launch": wkb * 1024.0 / launches, "bytes_per_launch": (2.0 * kb + wkb) * 1024.0 / launches}
json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps(out[workload], indent=1))
