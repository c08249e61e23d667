# round 6, sixteenth call: the ray kernels at 5 waves per SIMD - two workgroups of 640 threads per CU, 96 registers (k_trace 19 spilled, k_shadow_rays 16), each workgroup with its own
# LDS copy of the tree top: w5 = 8 stack entries per lane in LDS + 256 staged nodes, w5s6 = 6 entries + 336 nodes
out=gpurun_out/r06p; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  WORKLOADS="hall scan example" bash tools/gpu_ab_variants.sh $out/ab.txt default w5 w5s6
done
LUM_LIB=$PWD/luminary_amd/lib/variants/w5/libluminary_amd.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
