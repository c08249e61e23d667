# does the runtime's scratch policy cost the persistent ray kernels (1 KB of scratch per lane x 262 144 lanes = 281 MB per dispatch) a round trip per launch?
# an empty k_shadow_rays launch takes 72 us (profiles/r05_kernel_stats_hall.csv, MinNs). Kernel times per 3 steps under the runtime's scratch switches.
out=gpurun_out/r05u; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { echo "[$1]" | tee -a $out/ab.txt; WORKLOADS="hall example" bash tools/gpu_ab_variants.sh $out/ab.txt default; }
for rep in 1 2; do
  run "default"
  HSA_SCRATCH_SINGLE_LIMIT=2147483648 run "HSA_SCRATCH_SINGLE_LIMIT=2 GiB"
  HSA_NO_SCRATCH_RECLAIM=1 run "HSA_NO_SCRATCH_RECLAIM=1"
  HSA_SCRATCH_SINGLE_LIMIT_ASYNC=2147483648 run "HSA_SCRATCH_SINGLE_LIMIT_ASYNC=2 GiB"
done
