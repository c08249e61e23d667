# round 5: instruction-cache counters of the hall's kernels (one rocprofv3 --pmc pass, kernel trace only) and two more size-optimised builds
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $out/icache -- python3 bench.py --steps 2 --warmup 1 --cpu-budget 0 --secondary none > $out/icache.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = sorted(glob.glob(out + "/icache/**/*counter_collection.csv", recursive=True))[-1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].split("(")[0].split("::")[-1][:28]
    tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:6]:
    req = v.get("SQC_ICACHE_REQ", 0.0) or 1.0
    print("%-28s icache req %.3g  miss rate %.4f (dup %.4f)  ifetch %.3g  wait_inst_any / wave_cycles %.3f" % (k, req, v.get("SQC_ICACHE_MISSES", 0) / req, v.get("SQC_ICACHE_MISSES_DUPLICATE", 0) / req, v.get("SQ_IFETCH", 0), v.get("SQ_WAIT_INST_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1)))
PY
WORKLOADS="hall" bash tools/gpu_ab_variants.sh $out/ab_size.txt default opt_oz os_nounroll default
