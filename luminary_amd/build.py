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
HIP_SOURCES = ["host/core.hip", "host/lbvh.hip"]
EXTRA = os.environ.get("LUM_CXXFLAGS", "").split()
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


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


def build(force=False, verbose=False):
    if not force and not _newer(LIB, _all_sources()):
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs = []
    for s in HOST_SOURCES:
        o = os.path.join(OBJ_DIR, os.path.basename(s) + ".o")
        _run(["g++", *COMMON, "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROCM, "include"), "-c", os.path.join(CSRC, s), "-o", o])
        objs.append(o)
    bn = os.path.join(ROOT, "data", "bluenoise_2D.bin")
    o = os.path.join(OBJ_DIR, "embed.o")
    bn1 = os.path.join(ROOT, "data", "bluenoise_1D.bin")
    moon = [os.path.join(ROOT, "data", n) for n in ("moon_albedo.png", "moon_normal.png")]
    _run(["gcc", "-c", "-DLUM_BLUENOISE_PATH=\"%s\"" % bn, "-DLUM_BLUENOISE_1D_PATH=\"%s\"" % bn1, "-DLUM_MOON_ALBEDO_PATH=\"%s\"" % moon[0],
          "-DLUM_MOON_NORMAL_PATH=\"%s\"" % moon[1], os.path.join(CSRC, "host", "embed.S"), "-o", o])
    objs.append(o)
    for s in HIP_SOURCES:
        o = os.path.join(OBJ_DIR, os.path.basename(s) + ".o")
        # -fno-slp-vectorize: the SLP vectoriser packs neighbouring f32 multiplies/adds into v_pk_* and pays for it with register
        # moves (19 of the 61 instructions of a triangle test); same arithmetic, measured 2 % faster without it
        out = _run([HIPCC, "--offload-arch=gfx950", *COMMON, "-fno-slp-vectorize", *EXTRA, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, s), "-o", o])
        with open(os.path.join(OBJ_DIR, "kernel_resource_usage.txt"), "w" if s == HIP_SOURCES[0] else "a") as f:
            f.write(out)
        if verbose:
            print(out)
        objs.append(o)
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-lz", "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
