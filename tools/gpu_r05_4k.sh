# C4's frame on one GPU with the committed library (3840x2160, 32 sample ids per pass)
out=gpurun_out/r05s; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for w in hall example; do
  timeout 900 python bench.py --steps 3 --warmup 1 --cpu-budget 0 --secondary none --exact-steps 0 --width 3840 --height 2160 --workload $w 2>/dev/null | tail -1 > $out/bench_4k_$w.json
  python tools/ab_line.py < $out/bench_4k_$w.json | tee -a $out/ab.txt
done
