# round 6, eleventh call: k_shade at 4 waves per SIMD (128 registers) now that a tenth of its arithmetic is gone; k_shade's grid in rounds of its resident set at 64 ids per pass
out=gpurun_out/r06k; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  WORKLOADS="hall example" bash tools/gpu_ab_variants.sh $out/ab.txt default shade4w
done
WORKLOADS="hall example scan" bash tools/gpu_ab_env.sh "LUM_SHADE_GRID=1" "LUM_SHADE_GRID=2" "LUM_SHADE_GRID=3" "LUM_SHADE_GRID=4" | tee -a $out/ab.txt
