#!/usr/bin/env python3
"""One ad-hoc counter group on a bench workload, summed per kernel:  python tools/pmc_quick.py <out dir> <workload> COUNTER [COUNTER ...]
(separate from tools/pmc_collect.py's fixed passes; a group the hardware cannot take makes rocprofv3 hang, hence the time limit)"""
import csv
import glob
import os
import re
import signal
import subprocess
import sys
from collections import defaultdict

out, workload, counters = sys.argv[1], sys.argv[2], sys.argv[3:]
os.makedirs(out, exist_ok=True)
cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", out, "--", "python3", "bench.py", "--workload", workload, "--steps", "2", "--warmup", "1",
                                                           "--cpu-budget", "0", "--secondary", "none"]
p = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
try:
    p.wait(timeout=420)
except subprocess.TimeoutExpired:
    os.killpg(p.pid, signal.SIGKILL)
    sys.exit("rocprofv3 did not finish (counter group too large for the hardware?)")
sums = defaultdict(lambda: defaultdict(float))
launches = defaultdict(set)
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", re.sub(r"^void ", "", row["Kernel_Name"])).split("::")[-1]
        k = re.sub(r"<.*", "", k)
        sums[k][row["Counter_Name"]] += float(row["Counter_Value"])
        launches[k].add(row["Dispatch_Id"])
for k in sorted(sums, key=lambda k: -sum(sums[k].values())):
    print("%-22s launches %4d  " % (k, len(launches[k])) + "  ".join("%s %.4g" % (c, v) for c, v in sorted(sums[k].items())))
