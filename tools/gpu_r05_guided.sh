# the persistent ray kernels' reservations shrinking towards the end of the queue (LUM_GUIDED_CHUNKS, dev_trace.h) against fixed reservations of 256 rays (variant noguide)
out=gpurun_out/r05w; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  WORKLOADS="hall scan example" bash tools/gpu_ab_variants.sh $out/ab.txt noguide default
done
