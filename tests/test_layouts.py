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


def test_register_budgets_of_the_hot_kernels():
    """The occupancy the measurements in DESIGN.md section 4 (and docs/HISTORY.md) rest on, read from the compiler's resource report of the last build
    (luminary_amd/lib/obj/kernel_resource_usage.txt): the fast flavour's ray kernels fit 128 registers (4 waves per SIMD, one 1024-thread workgroup per
    CU) without spilling (since round 6's two-triangle leaves), k_shade<constant sky> runs at 3 waves, k_clouds at 4, and no shading kernel falls to a single wave."""
    import subprocess
    path = os.path.join(ROOT, "luminary_amd", "lib", "obj", "kernel_resource_usage.txt")
    assert os.path.exists(path), "run `python -m luminary_amd.build` first"
    rows, cur = [], None
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, short in (("VGPRs", "vgpr"), ("Occupancy [waves/SIMD]", "occ"), ("VGPRs Spill", "spill")):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m:
                cur[short] = int(m.group(1))
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
    table = {re.sub(r"\(.*", "", n).replace("void ", ""): r for r, n in zip(rows, names)}
    # round 6: leaves of at most two triangles (LUM_LEAF_MAX) took 24 registers out of the ray kernels - nothing is spilled any more (k_trace: 128 + 12 spilled before)
    for k in ("lum::fast::k_trace", "lum::fast::k_shadow_rays", "lum::fast::k_trace_particles", "lum::exact::k_trace", "lum::exact::k_shadow_rays"):
        assert table[k]["occ"] >= 4 and table[k]["vgpr"] <= 120 and table[k]["spill"] == 0, (k, table[k])
    for with_table in ("true", "false"):  # constant sky, no ocean, the whole vertex in one kernel (stage 0: the product); with the pass's Sobol table and hashing
        shade = table["lum::fast::k_shade<2u, false, 0, %s>" % with_table]
        assert shade["occ"] == 3 and shade["spill"] <= 12, shade  # (10 with the input cursor's four wave-uniform words; measured faster all the same)
    assert table["lum::fast::k_clouds"]["occ"] == 4
    for k, r in table.items():
        if k.startswith("lum::fast::k_") and "occ" in r:
            assert r["occ"] >= 2, (k, r)
    for k in ("lum::fast::k_particle_shade", "lum::fast::k_ocean_shade", "lum::fast::k_volume_inscatter"):
        assert table[k]["occ"] >= 3, (k, table[k])
