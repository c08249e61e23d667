"""The two arithmetic flavours of the device code (luminary_amd/csrc/device/flavour.h, include/lum_core.h lumc_set_flavour).

exact = what every parity test runs: bit-identical to the oracle. fast = the product's default (hardware reciprocal / rsqrt / sqrt / sin /
cos / exp2 / log2, fused multiply-add: how the reference itself is built, --use_fast_math, src/luminary/CMakeLists.txt:48). This module
gates fast against exact with the north star's tolerance: relative L2 of the radiance < 1e-3 at 1024 spp with identical sample ids, and
ray counters within 0.1 %."""
import numpy as np
import pytest

import oracle_lib
from luminary_amd import scenes
from luminary_amd.core import Core


def test_flavour_api_without_gpu_is_declared():
    """The C ABI exports the flavour switch (symbol check only, no GPU needed)."""
    from luminary_amd import _lib
    lib = _lib()
    assert hasattr(lib, "lumc_set_flavour") and hasattr(lib, "lumc_get_flavour")


def _render(core, view, flavour, spp, batch):
    core.set_flavour(flavour)
    assert core.flavour == flavour
    core.set_pixels(None)
    core.reset_counters()
    core.render(0, spp, samples_per_pass=batch)
    fm, sm = core.accumulators()
    return fm / np.float32(spp), core.counters()


def _rel_l2(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["cornell", "zoo"])
def test_fast_flavour_within_tolerance_of_exact(scene, tmp_path_factory):
    if scene == "cornell":
        host = scenes.cornell_host(str(tmp_path_factory.mktemp("cornell")), 64, 64, 8)
    else:
        host = scenes.zoo_scene(96, 64, 8)
    view = oracle_lib.with_luts(host.device_scene())
    core = Core(0)
    try:
        core.upload(view)
        exact, cnt_exact = _render(core, view, "exact", 1024, 64)
        fast, cnt_fast = _render(core, view, "fast", 1024, 64)
        again, _ = _render(core, view, "exact", 1024, 64)
    finally:
        core.close()
    assert np.array_equal(exact, again), "switching flavours back and forth must not change the exact result"
    assert np.isfinite(fast).all()
    err = _rel_l2(fast, exact)
    assert err < 1e-3, "relative L2 radiance error of the fast flavour at 1024 spp: %g" % err
    for k, name in enumerate(("closest-hit rays", "shadow rays", "light-BVH queries", "vertices")):
        assert abs(cnt_fast[k] - cnt_exact[k]) <= 1e-3 * max(cnt_exact[k], 1), "%s differ by more than 0.1 %%: %d vs %d" % (name, cnt_fast[k], cnt_exact[k])
    assert not np.array_equal(fast, exact), "the fast flavour is expected to differ in the last bits (otherwise it is not being run)"


@pytest.mark.gpu
def test_default_flavour_is_fast_and_env_selects_exact(monkeypatch):
    """lumc_context_create: fast unless LUM_FLAVOUR says otherwise (the test session sets LUM_FLAVOUR=exact, tests/conftest.py)."""
    c = Core(0)
    assert c.flavour == "exact"
    c.close()
    monkeypatch.delenv("LUM_FLAVOUR")
    c = Core(0)
    assert c.flavour == "fast"
    c.close()
