# three more compiler switches for all HIP sources: -amdgpu-use-amdgpu-trackers, -amdgpu-schedule-relaxed-occupancy, -amdgpu-max-memory-clause=4
out=gpurun_out/r05x; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  WORKLOADS="hall example" bash tools/gpu_ab_variants.sh $out/ab.txt default trk relax clause4
done
