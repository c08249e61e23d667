"""The pass's Sobol table (include/lum_core.h lumc_set_sobol_table; csrc/device/dev_sampler.h SamplerT<kTable>; k_sobol_table in kernels.h).

The Sobol / Owen part of a random number (cuda/random.cuh:238-287) is a function of (sample id, dimension) alone. lumc_render writes it once per pass for the
pass's consecutive sample ids and every dimension the depth loop can ask for, and the shading kernel's k_shade<..., kTable = true> instances read it instead of
hashing. The integers are the same, so a frame must not change by a bit - in either flavour, for any first sample id, ragged last passes, pixel subsets, depth 0,
and passes too large for a table (which hash). Every oracle-parity test of the suite renders through lumc_render in the exact flavour, i.e. through the table."""
import numpy as np
import pytest

import oracle_lib
from luminary_amd import SKY_MODE_DEFAULT, scenes
from luminary_amd.core import Core


def test_the_c_abi_exports_the_switch():
    from luminary_amd import _lib
    assert hasattr(_lib(), "lumc_set_sobol_table")


def _frame(core, on, first, count, batch, pixels=None):
    core.set_sobol_table(on)
    core.set_pixels(pixels)
    core.reset_counters()
    core.render(first, count, samples_per_pass=batch)
    fm, sm = core.accumulators()
    return fm.copy(), sm.copy(), list(core.counters())


def _host(name, tmp):
    if name == "cornell":
        return scenes.cornell_host(str(tmp), 64, 64, 5)
    if name == "zoo":        # hundreds of emitters: the light tree's descent draws from the postpass targets
        return scenes.zoo_scene(96, 64, 8)
    if name == "example":
        return scenes.example_scene(128, 80, 6, sphere_segments=8, ground_res=12, num_objects=12, num_lights=4)
    if name == "sky":        # procedural sky: the sun sample's targets, k_shade<default sky>
        return scenes.zoo_scene(64, 48, 4, sky_mode=SKY_MODE_DEFAULT)
    if name == "ocean":      # k_shade<.., water>: caustics' targets
        host = scenes.zoo_scene(64, 48, 4, sky_mode=SKY_MODE_DEFAULT)
        o = host.get_ocean()
        o.active, o.height, o.amplitude, o.frequency = True, 0.4, 0.2, 0.5
        host.set_ocean(o)
        return host
    raise ValueError(name)


@pytest.mark.gpu
@pytest.mark.parametrize("flavour", ["exact", "fast"])
@pytest.mark.parametrize("name", ["cornell", "zoo", "example", "sky", "ocean"])
def test_a_frame_is_the_same_bits_with_the_table_and_without(name, flavour, tmp_path):
    host = _host(name, tmp_path)
    view = oracle_lib.with_luts(host.device_scene())
    core = Core(0)
    try:
        core.set_flavour(flavour)
        core.upload(view)
        # first id 5, eleven ids in passes of 4: passes of 4, 4 and 3 ids starting at 5, 9 and 13
        fm0, sm0, c0 = _frame(core, False, 5, 11, 4)
        fm1, sm1, c1 = _frame(core, True, 5, 11, 4)
    finally:
        core.close()
    assert np.isfinite(fm1).all() and float(fm1.sum()) > 0.0
    assert c1 == c0
    assert np.array_equal(fm1, fm0) and np.array_equal(sm1, sm0)


@pytest.mark.gpu
def test_large_passes_pixel_subsets_and_depth_zero(tmp_path):
    """A pass of more ids than a table is built for hashes (same bits); a pass of exactly the limit has one; a pixel subset and max_ray_depth = 0 (one row of
    dimensions) work; a table of another size follows the passes (its buffer grows with the pass)."""
    host = scenes.example_scene(32, 24, 3, sphere_segments=6, ground_res=6, num_objects=6, num_lights=3)
    view = oracle_lib.with_luts(host.device_scene())
    core = Core(0)
    try:
        core.upload(view)   # the suite's flavour: exact
        for first, count, batch in [(0, 1100, 1100), (7, 1024, 1024), (0, 300, 300), (1000, 40, 16), (3, 2, 1)]:
            fm0, sm0, c0 = _frame(core, False, first, count, batch)
            fm1, sm1, c1 = _frame(core, True, first, count, batch)
            assert c1 == c0 and np.array_equal(fm1, fm0) and np.array_equal(sm1, sm0), (first, count, batch)
        subset = np.arange(5, 32 * 24, 7, dtype=np.uint32)
        fm0, _, _ = _frame(core, False, 2, 6, 3, subset)
        fm1, _, _ = _frame(core, True, 2, 6, 3, subset)
        assert np.array_equal(fm1, fm0)
    finally:
        core.close()
    host0 = scenes.example_scene(32, 24, 0, sphere_segments=6, ground_res=6, num_objects=6, num_lights=3)
    core = Core(0)
    try:
        core.upload(oracle_lib.with_luts(host0.device_scene()))
        fm0, _, _ = _frame(core, False, 0, 8, 4)
        fm1, _, _ = _frame(core, True, 0, 8, 4)
    finally:
        core.close()
    assert np.array_equal(fm1, fm0) and float(fm1.sum()) > 0.0
