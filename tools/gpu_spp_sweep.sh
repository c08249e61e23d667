#!/bin/bash
# Sample ids per wavefront pass, 32 / 64 / 128, on the three workloads (GPU box): bash tools/gpu_spp_sweep.sh > gpurun_out/spp_sweep.txt
for w in hall example scan; do for n in 32 64 128; do echo -n "[$w --samples-per-pass $n] "; python bench.py --steps 3 --warmup 1 --cpu-budget 0 --secondary none --workload $w --samples-per-pass $n 2>/dev/null | python tools/ab_line.py; done; done
