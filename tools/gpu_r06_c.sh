# round 6, third call: the GPU suite of the tree with two-triangle leaves + the optimal collapse on the host and on the GPU, then the new defaults (64 ids) on the three workloads,
# the greedy rule through the GPU builder for comparison, and the builders' table
out=gpurun_out/r06c; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $out/pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for rep in 1 2; do
  WORKLOADS="hall scan example" bash tools/gpu_ab_env.sh "LUM_BVH_COLLAPSE=1" "LUM_BVH_COLLAPSE=0" "LUM_BVH_BUILDER=sah" | tee -a $out/ab.txt
done
timeout 600 python tools/lbvh_bench.py hall > $out/lbvh_hall.txt 2>&1; tail -6 $out/lbvh_hall.txt
timeout 600 python tools/lbvh_bench.py scan > $out/lbvh_scan.txt 2>&1; tail -6 $out/lbvh_scan.txt
