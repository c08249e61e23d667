"""Generates tests/golden/bsdf_luts.npz with the ORACLE's BSDF energy tables (reference algorithm: cuda/bsdf_lut.cuh:20-211).

Run from the repo root: python tests/golden/make_luts.py   (about 3-4 minutes on 8 cores; output ~130 KB)
The fixture is data: four u16 tables (conductor 32x32, glossy 32x32, dielectric 32^3, dielectric_inv 32^3).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
lib = C.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
bn = np.fromfile(os.path.join(ROOT, "luminary_amd", "data", "bluenoise_2D.bin"), dtype=np.uint32)
assert bn.size == 65536


def table(idx, count, conductor=None):
    out = np.zeros(count, dtype=np.uint16)
    cp = conductor.ctypes.data_as(C.c_void_p) if conductor is not None else C.c_void_p(0)
    rc = lib.oracle_generate_lut(bn.ctypes.data_as(C.c_void_p), C.c_int(idx), C.c_uint32(0), C.c_uint32(count), cp, out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out


conductor = table(0, 1024)
glossy = table(1, 1024, conductor)
dielectric = table(2, 32768)
dielectric_inv = table(3, 32768)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bsdf_luts.npz"), conductor=conductor, glossy=glossy, dielectric=dielectric,
                    dielectric_inv=dielectric_inv)
print("ok", conductor[:4], glossy[:4], dielectric[:4], dielectric_inv[:4])
