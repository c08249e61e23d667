#!/bin/sh
# Copies what tools/gpu_round_profile.sh left under gpurun_out/<tag>/ into profiles/ (the tracked evidence): bench lines, rocprofv3 kernel
# statistics, PMC summaries, the GPU test log, the tools' output.   sh tools/collect_round_evidence.sh r01
set -e
tag=${1:-r01}
o=gpurun_out/$tag
python tools/pmc_traffic.py example 8 $o/pmc_fetch_example $o/pmc_write_example > /dev/null
python tools/pmc_traffic.py hall 8 $o/pmc_fetch_hall $o/pmc_write_hall > /dev/null
for w in example hall scan; do grep '^{' $o/bench_$w.json | tail -1 > profiles/${tag}_bench_$w.json; done
cp $o/stats_example/*/*_kernel_stats.csv profiles/${tag}_kernel_stats_example.csv
cp $o/stats_hall/*/*_kernel_stats.csv profiles/${tag}_kernel_stats_hall.csv
cp $o/pytest_gpu.log profiles/${tag}_pytest_gpu.log
{ echo "# SQ LDS counters, C2 (python bench.py --steps 2 --warmup 1): bank-conflict cycles vs LDS-active cycles of the ray kernels"; python tools/pmc_summary.py $o/pmc_lds_example
  echo; echo "# TCC counters, C3 hall: L2 hits / misses, memory-side read requests and those that reached DRAM"; python tools/pmc_summary.py $o/pmc_tcc_hall; } > profiles/${tag}_pmc_lds_tcc_summary.txt
{ echo "## tools/lbvh_bench.py hall"; cat $o/lbvh_hall.txt; echo; echo "## tools/adaptive_bench.py example 8 2"; cat $o/adaptive_example.txt; echo
  echo "## tools/phase_stats.py example (diagnostic build -DLUM_PHASE_STATS)"; cat $o/phase_example.txt; echo; echo "## tools/phase_stats.py hall"; cat $o/phase_hall.txt; } > profiles/${tag}_tools_output.txt
python - <<PY
import json
for w in ("example", "hall", "scan"):
    d = json.loads(open("profiles/${tag}_bench_%s.json" % w).read())
    print(w, round(d["value"]), "Mrays/s,", round(d["ms_per_step"], 1), "ms/step, roofline.frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
PY
tail -1 profiles/${tag}_pytest_gpu.log
