# round 5: phase-queue variants - parity of the exact flavour, then the A/B bench (bash tools/gpu_r05_pq.sh <out dir> <variants...>)
out=$1; shift
mkdir -p $out
first=$1
LUM_LIB=$PWD/luminary_amd/lib/variants/$first/libluminary_amd.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_lbvh.py tests/test_ambient_reuse.py -m gpu -x -q > $out/parity_$first.log 2>&1
tail -3 $out/parity_$first.log
WORKLOADS="${WORKLOADS:-hall scan example}" bash tools/gpu_ab_variants.sh $out/ab.txt default "$@"
