# A/B of library variants built here (python -m luminary_amd.build --variant NAME with LUM_CXXFLAGS / LUM_FAST_FLAGS set) on the GPU box:
#   bash tools/gpu_ab_variants.sh <out file> default NAME ...     workloads in $WORKLOADS (default "hall scan"), extra bench arguments in $BENCH_ARGS
out=$1; shift
mkdir -p $(dirname $out)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "$@"; do
  lib=$PWD/luminary_amd/lib/variants/$v/libluminary_amd.so
  [ "$v" = default ] && lib=$PWD/luminary_amd/lib/libluminary_amd.so
  [ -f $lib ] || { echo "[$v] no library" | tee -a $out; continue; }
  for w in ${WORKLOADS:-hall scan}; do
    echo -n "[$v] $w $BENCH_ARGS: " | tee -a $out
    LUM_LIB=$lib timeout 600 python bench.py --steps 3 --warmup 1 --cpu-budget 0 --secondary none --workload $w $BENCH_ARGS 2>/dev/null | python tools/ab_line.py | tee -a $out
  done
done
