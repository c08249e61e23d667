run() { python bench.py --steps 3 --warmup 1 --cpu-budget 0 --secondary none --workload hall 2>/dev/null | python tools/ab_line.py; }
NOAPPROX="-DLUM_FAST=1 -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -freciprocal-math -fno-math-errno -fgpu-flush-denormals-to-zero"
export LUM_CXXFLAGS="-DLUM_BVH8=0"
python -m luminary_amd.build --force > /dev/null 2>&1; echo -n "[bvh4 default 1] "; run
echo -n "[bvh4 default 2] "; run
LUM_FAST_FLAGS="$NOAPPROX" python -m luminary_amd.build --force > /dev/null 2>&1; echo -n "[bvh4 no-approx-func 1] "; LUM_FAST_FLAGS="$NOAPPROX" run
echo -n "[bvh4 no-approx-func 2] "; LUM_FAST_FLAGS="$NOAPPROX" run
python -m luminary_amd.build --force > /dev/null 2>&1; echo -n "[bvh4 default 3] "; run
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
