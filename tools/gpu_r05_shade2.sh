# k_shade follow-ups to the Sobol table: flat = the light tree's descent compiled out (LUM_ABLATE_DESCENT: scenes with <= 128 lights only), bns = the blue-noise
# offsets formed on the scalar unit at the use (LUM_BLUENOISE_SALU); then C4's frame on one GPU with the committed library
out=gpurun_out/r05o; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  WORKLOADS="hall scan example" bash tools/gpu_ab_variants.sh $out/ab.txt default flat bns
done
for w in hall example; do
  timeout 900 python bench.py --steps 3 --warmup 1 --cpu-budget 0 --secondary none --exact-steps 0 --width 3840 --height 2160 --workload $w 2>/dev/null | tail -1 > $out/bench_4k_$w.json
  python tools/ab_line.py < $out/bench_4k_$w.json | tee -a $out/ab.txt
done
