# Round-6 evidence of the committed tree, one gpurun call: the GPU suite + smoke, the PMC passes (before the bench line: its roofline blocks carry counter figures only
# for the source tree they were collected from), the default bench line, rocprofv3 kernel statistics of the same command, the flavour gate on the north-star scene, and
# C4's frame (3840x2160) on one GPU.   gpurun --timeout 3300 -- 'bash tools/gpu_r06_evidence.sh'   then   bash tools/refresh_profiles.sh r06 r06
bash tools/gpu_round2.sh r06 tests pmc bench stats
out=gpurun_out/r06
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python tools/flavour_gate.py > $out/flavour_gate_r06.json 2> $out/flavour_gate_r06.err; tail -c 700 $out/flavour_gate_r06.json
for w in hall example; do
  timeout 900 python bench.py --steps 3 --warmup 1 --cpu-budget 0 --secondary none --exact-steps 0 --width 3840 --height 2160 --samples-per-pass 32 --workload $w 2>/dev/null | tail -1 > $out/bench_4k_$w.json
  python tools/ab_line.py < $out/bench_4k_$w.json
done
