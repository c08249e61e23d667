#!/bin/sh
# The host layer under AddressSanitizer + UBSan on the CPU (GPU AddressSanitizer is not available on the pool): the six host C++ files rebuilt with
# -fsanitize=address,undefined, linked with the product's own HIP objects, and the CPU test suite run against that library (LUM_LIB).
#   python -m luminary_amd.build && sh tools/asan_cpu.sh          round 4: 152 CPU tests, no report; round 6: 202 (incl. the collapse plan and its exhaustive check), no report
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/lum_asan
mkdir -p $OUT
for s in scene bvh_build loaders api utils_api output; do
  g++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fno-fast-math -D__HIP_PLATFORM_AMD__ -I ${ROCM_PATH:-/opt/rocm}/include \
      -c $ROOT/luminary_amd/csrc/host/$s.cpp -o $OUT/$s.o &
done
wait
cp $ROOT/luminary_amd/lib/obj/embed.o $ROOT/luminary_amd/lib/obj/core.hip.o $ROOT/luminary_amd/lib/obj/lbvh.hip.o $ROOT/luminary_amd/lib/obj/wavefront_fast.hip.o $ROOT/luminary_amd/lib/obj/wavefront_fast_shadow.hip.o $ROOT/luminary_amd/lib/obj/wavefront_exact_shadow.hip.o $OUT/
${ROCM_PATH:-/opt/rocm}/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined $OUT/*.o -lz -L ${ROCM_PATH:-/opt/rocm}/lib -lrccl -o $OUT/libluminary_amd.so
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 LUM_LIB=$OUT/libluminary_amd.so \
  python -m pytest $ROOT/tests -x -q -m "not gpu" --deselect $ROOT/tests/test_distributed_cpu.py
