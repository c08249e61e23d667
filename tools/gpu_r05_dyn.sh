# k_shade's input through a cursor (LUM_SHADE_DYNAMIC, kernels.h) against the fixed shares (variant `static`), and the grid that goes with it (LUM_SHADE_GRID rounds)
out=gpurun_out/r05q; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ "$1" = parity ]; then timeout 1500 python -m pytest tests/test_sobol_table.py tests/test_gpu_parity.py tests/test_ambient_reuse.py tests/test_flavours.py -m gpu -x -q 2>&1 | tail -3 | tee $out/parity.txt; fi
for rep in 1 2; do
  WORKLOADS="hall scan example" bash tools/gpu_ab_variants.sh $out/ab3.txt static
  for g in 1 2 3 4; do
    echo "[guided + early exit, LUM_SHADE_GRID=$g]" | tee -a $out/ab3.txt
    LUM_SHADE_GRID=$g WORKLOADS="hall scan example" bash tools/gpu_ab_variants.sh $out/ab3.txt default
  done
done
