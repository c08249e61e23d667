#!/bin/bash
# What the widened features cost on the Example-class scene (fast flavour, 1920x1080, 8 bounces, 8 sample ids per step): kernel ms per run.
# Usage (GPU box): bash tools/gpu_feature_cost.sh > gpurun_out/feature_cost.txt
B="python bench.py --workload example --secondary none --cpu-budget 0 --steps 3 --warmup 1 --samples-per-pass 8"
sky() { python -c "import json,sys,os; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); d=json.load(open(d[\"detail\"])) if \"kernel_ms_rank0\" not in d[\"config\"] and d.get(\"detail\") else d; k=d[\"config\"][\"kernel_ms_rank0\"]; print(\"   sky group %.1f ms, volume group %.1f ms, total %.1f ms/step\" % (k.get(\"sky\",0.0), k.get(\"volume\",0.0), d[\"ms_per_step\"]))"; }
for cfg in "--sky constant" "--sky procedural" "--sky procedural --fog 40" "--sky procedural --ocean 0.5" "--sky procedural --ocean 6" "--clouds" "--clouds --fog 40 --ocean 0.5"; do
  echo "[$cfg]"
  $B $cfg 2>/dev/null | tail -1 > /tmp/fc_line.json
  python tools/ab_line.py < /tmp/fc_line.json
  sky < /tmp/fc_line.json
done
