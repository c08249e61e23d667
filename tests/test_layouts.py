"""The C-ABI library loads, exports every symbol its headers declare, and shares the scene layout with the oracle. No GPU needed."""
import ctypes as C
import os
import re

import oracle_lib
import luminary_amd
from luminary_amd import core

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#define.*?(?<!\\)$", "", text, flags=re.M | re.S)  # macros wrap the underscore-prefixed entry points
    return sorted(set(re.findall(r"LUMINARY_API[^;(]*?\b([a-z_][a-z0-9_]+)\s*\(", text)) | set(re.findall(r"\b(lumc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = luminary_amd._lib()
    names = _declared_functions("luminary_amd.h") + _declared_functions("lum_core.h")
    assert len(names) > 110 and "_array_push" in names and "queue_pop_blocking" in names and "thread_status_start" in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_scene_view_layout_is_shared():
    assert core.scene_view_sizeof() == C.sizeof(luminary_amd.DeviceSceneView) == oracle_lib.lib().oracle_scene_sizeof()


def test_api_struct_sizes_match_the_c_header():
    # compiled probe of include/luminary_amd.h
    import subprocess
    import tempfile
    src = '#include "luminary_amd.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(LuminaryRendererSettings), sizeof(LuminaryCamera), sizeof(LuminarySky), sizeof(LuminaryMaterial), sizeof(LuminaryInstance));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), os.path.join(d, "p.c"), "-o", os.path.join(d, "p")])
        got = [int(x) for x in subprocess.check_output([os.path.join(d, "p")]).split()]
    want = [C.sizeof(luminary_amd.RendererSettings), C.sizeof(luminary_amd.Camera), C.sizeof(luminary_amd.Sky), C.sizeof(luminary_amd.Material),
            C.sizeof(luminary_amd.Instance)]
    assert got == want


def test_rendering_without_gpu_fails_loudly():
    import torch
    if torch.cuda.device_count() > 0:
        return
    try:
        core.Core(0)
    except core.CoreError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("Core() must not succeed without a HIP device")
