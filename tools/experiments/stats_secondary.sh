# rocprofv3 kernel stats of the two secondary workloads (the headline's are tools/gpu_round2.sh stats): bash tools/experiments/stats_secondary.sh <tag>
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for w in example scan; do
  mkdir -p gpurun_out/$tag/stats_$w
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/stats_$w -- python3 bench.py --workload $w --steps 4 --warmup 1 --cpu-budget 0 --secondary none > gpurun_out/$tag/stats_$w.log 2>&1
  find gpurun_out/$tag/stats_$w -name "*kernel_stats.csv" | head -1 | xargs head -8 | cut -c1-200
done
