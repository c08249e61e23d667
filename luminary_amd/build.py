"""Builds libluminary_amd.so (HIP kernels for gfx950 + C-ABI host layer) in-tree.

Usage: python -m luminary_amd.build [--force]
Host C++ is compiled with g++, the kernels with hipcc --offload-arch=gfx950, everything is linked by hipcc.
-ffp-contract=off is part of the numerics contract (DESIGN.md "Determinism").
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "csrc")
LIB_DIR = os.path.join(ROOT, "lib")
LIB = os.path.join(LIB_DIR, "libluminary_amd.so")
OBJ_DIR = os.path.join(ROOT, "lib", "obj")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")

HOST_SOURCES = ["host/scene.cpp", "host/bvh_build.cpp", "host/loaders.cpp", "host/api.cpp", "host/utils_api.cpp", "host/output.cpp"]
EXTRA = os.environ.get("LUM_CXXFLAGS", "").split()
COMMON = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# The HIP sources are compiled with -Os. Measured (profiles/r05_ab_experiments.txt, three interleaved repeats): 1.3 % off k_trace, 0.7 % off k_shade, 1.5 % off
# k_shadow_rays = +1.0 % samples/s on the hall (scan +0.6 %, Example-class +0.8 %); -O2 half of that, -Oz loses 7 % in k_shade. Not an instruction-cache effect
# (SQC_ICACHE_MISSES / SQC_ICACHE_REQ = 0.0000 for all three kernels under -O3): -Os simply emits fewer instructions on the executed paths (less speculation and
# duplication of code around branches). It comes after -O3 on the command line and wins.
HIP_OPT = os.environ.get("LUM_HIP_OPT", "-Os").split()
EXACT = ["-ffp-contract=off", "-fno-fast-math"]  # the numerics contract of the exact flavour and of all host code
# the fast flavour of the wavefront kernels (csrc/device/flavour.h): contraction, hardware reciprocal / sqrt, reciprocal-multiply for x / y.
# -fapprox-func: without it 28 divisions of k_shade stay correctly rounded (v_div_scale / v_div_fmas / v_div_fixup with two denormal-mode
# switches each) although the 2.5-ulp division is allowed - the reciprocals -freciprocal-math creates lose that permission
FAST = ["-DLUM_FAST=1", "-ffp-contract=fast", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-freciprocal-math", "-fno-math-errno", "-fapprox-func",
        "-fgpu-flush-denormals-to-zero"]  # denormals flushed like the reference's --use_fast_math build: a/b is v_rcp + v_mul, sqrt is v_sqrt (measured +1 %)
if os.environ.get("LUM_FAST_FLAGS") is not None:  # diagnosis only (tools/flavour_diff.py): which part of the fast flavour moves the image
    FAST = ["-DLUM_FAST=1"] + os.environ["LUM_FAST_FLAGS"].split()
# -fno-slp-vectorize: the SLP vectoriser packs neighbouring f32 multiplies/adds into v_pk_* and pays for it with register
# moves (19 of the 61 instructions of a triangle test); same arithmetic, measured 2 % faster without it
# each flavour's visibility-ray kernel is a translation unit of its own: the max-ILP scheduler takes 3-4 % off it and costs the other kernels 1-2 %
# (csrc/device/kernel_shadow.h, profiles/r05_ab_experiments.txt); LUM_FAST_SHADOW_SCHED= (empty) in the environment builds it with the default scheduler
SHADOW_SCHED = os.environ.get("LUM_FAST_SHADOW_SCHED", "max-ilp")
SHADOW_FLAGS = (["-mllvm", "-amdgpu-sched-strategy=" + SHADOW_SCHED] if SHADOW_SCHED else [])
HIP_SOURCES = [("host/core.hip", EXACT + ["-DLUM_SHADOW_KERNEL_EXTERN=1"]), ("host/lbvh.hip", EXACT), ("device/wavefront_fast.hip", FAST + ["-DLUM_SHADOW_KERNEL_EXTERN=1"]),
               ("device/wavefront_fast_shadow.hip", FAST + SHADOW_FLAGS), ("device/wavefront_exact_shadow.hip", EXACT + SHADOW_FLAGS)]
STAMP = os.path.join(LIB_DIR, "build_flags.txt")


def _flags_identity():
    return "\n".join(["common " + " ".join(COMMON), "hip " + " ".join(HIP_OPT), "exact " + " ".join(EXACT), "fast " + " ".join(FAST), "shadow " + " ".join(SHADOW_FLAGS), "extra " + " ".join(EXTRA)]) + "\n"


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _all_sources():
    out = []
    for d, _, files in os.walk(CSRC):
        out += [os.path.join(d, f) for f in files]
    out.append(os.path.join(ROOT, "..", "include", "lum_core.h"))
    out.append(os.path.join(ROOT, "..", "include", "luminary_amd.h"))
    out.append(os.path.abspath(__file__))
    return out


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build step failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def build(force=False, verbose=False, variant=None):
    """variant: an experiment build (LUM_CXXFLAGS / LUM_FAST_FLAGS of the calling environment) into lib/variants/<variant>/ next to the default
    library, compiled here and selected on the GPU box with LUM_LIB=<path>: an A/B run then costs no compile time there."""
    global LIB, OBJ_DIR, STAMP
    if variant:
        vdir = os.path.join(LIB_DIR, "variants", variant)
        LIB, OBJ_DIR, STAMP = os.path.join(vdir, "libluminary_amd.so"), os.path.join(vdir, "obj"), os.path.join(vdir, "build_flags.txt")
    # the effective flag list is part of the build's identity: a library left behind by a diagnostic build (LUM_CXXFLAGS=-DLUM_PHASE_STATS,
    # an ablation ...) is rebuilt instead of being silently reused
    stamp_ok = os.path.exists(STAMP) and open(STAMP).read() == _flags_identity()
    if not force and stamp_ok and not _newer(LIB, _all_sources()):
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    if os.path.exists(STAMP):
        os.remove(STAMP)
    objs = []
    for s in HOST_SOURCES:
        o = os.path.join(OBJ_DIR, os.path.basename(s) + ".o")
        # (the -D switches of an experiment reach the host sources too: constants like LUM_LEAF_MAX are shared with the builders there)
        _run(["g++", *COMMON, *EXACT, *[f for f in EXTRA if f.startswith("-D")], "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROCM, "include"), "-c", os.path.join(CSRC, s), "-o", o])
        objs.append(o)
    bn = os.path.join(ROOT, "data", "bluenoise_2D.bin")
    o = os.path.join(OBJ_DIR, "embed.o")
    bn1 = os.path.join(ROOT, "data", "bluenoise_1D.bin")
    moon = [os.path.join(ROOT, "data", n) for n in ("moon_albedo.png", "moon_normal.png")]
    _run(["gcc", "-c", "-DLUM_BLUENOISE_PATH=\"%s\"" % bn, "-DLUM_BLUENOISE_1D_PATH=\"%s\"" % bn1, "-DLUM_MOON_ALBEDO_PATH=\"%s\"" % moon[0],
          "-DLUM_MOON_NORMAL_PATH=\"%s\"" % moon[1], "-DLUM_BRIDGE_LUT_PATH=\"%s\"" % os.path.join(ROOT, "data", "bridge_lut.bin"),
          os.path.join(CSRC, "host", "embed.S"), "-o", o])
    objs.append(o)
    from concurrent.futures import ThreadPoolExecutor

    def compile_hip(item):
        src, flags = item
        o = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        out = _run([HIPCC, "--offload-arch=gfx950", *COMMON, *HIP_OPT, *flags, "-fno-slp-vectorize", *EXTRA, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", o])
        return o, out

    with ThreadPoolExecutor(max_workers=len(HIP_SOURCES)) as pool:
        results = list(pool.map(compile_hip, HIP_SOURCES))
    with open(os.path.join(OBJ_DIR, "kernel_resource_usage.txt"), "w") as f:
        for (src, _), (o, out) in zip(HIP_SOURCES, results):
            f.write("#### %s\n%s" % (src, out))
            if verbose:
                print(out)
            objs.append(o)
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-lz", "-L", os.path.join(ROCM, "lib"), "-lrccl", "-o", LIB])
    with open(STAMP, "w") as f:
        f.write(_flags_identity())
    return LIB


if __name__ == "__main__":
    variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, variant=variant))
