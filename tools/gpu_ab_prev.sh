#!/bin/bash
# Same-box A/B of the working tree's kernels.h against the version saved in tools/experiments/kernels_prev.h.txt (bash tools/gpu_ab_prev.sh "<bench args>" ...)
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['config']['kernel_ms_rank0']; print('%.1f Mrays/s | trace %.1f shade %.1f shadow %.1f lq %.1f res %.1f sky %.1f vol %.1f | %.1f ms/step' % (d['value'], k['trace'], k['shade'], k['shadow'], k['light_query'], k['resolve'], k.get('sky',0.0), k.get('volume',0.0), d['ms_per_step']))"; }
K=luminary_amd/csrc/device/kernels.h
cp $K /tmp/kernels_new.h
trap 'cp /tmp/kernels_new.h $K; python -m luminary_amd.build --force > /dev/null 2>&1' EXIT
for rep in 1 2; do
for v in prev new; do
  if [ $v = prev ]; then cp tools/experiments/kernels_prev.h.txt $K; else cp /tmp/kernels_new.h $K; fi
  python -m luminary_amd.build --force > /dev/null 2>&1 || { echo "[$v] build failed"; continue; }
  for cfg in "$@"; do
    echo -n "[$v] $cfg: "
    python bench.py --secondary none --cpu-budget 0 --steps 3 --warmup 1 $cfg 2>/dev/null | line
  done
done
done
