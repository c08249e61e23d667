"""Adaptive sampling: stage schedule, per-block rates, per-pixel sample ids and the result image.
CPU part: properties of the oracle restatement. GPU part: lumc_adaptive_* / lumc_generate_result against the oracle, bit for bit."""
import numpy as np
import pytest

import oracle_lib
from luminary_amd import scenes
from luminary_amd.core import Core, default_output_params

W, H, BOUNCES = 30, 22, 3          # partial 4x4 blocks on both edges
MAX_RATE, AVG_RATE, INTERVAL = 6, 2, 2


def _scene(tmp_path_factory):
    d = tmp_path_factory.mktemp("cornell_adaptive")
    host = scenes.cornell_host(str(d), W, H, BOUNCES)
    return oracle_lib.with_luts(host.device_scene())


def _oracle_run(view, executions, exposure):
    tone = default_output_params(W, H, 1)
    o = oracle_lib.AdaptiveOracle(view, MAX_RATE, AVG_RATE, INTERVAL, exposure=exposure, tone=tone)
    o.render(executions)
    return o, tone


def test_oracle_stage_schedule_and_rates(tmp_path_factory):
    view = _scene(tmp_path_factory)
    # stage 0 lasts 2 executions, stage 1 lasts 4, stage 2 lasts 8: after 2 + 4 + 3 executions we are in stage 2
    o, _ = _oracle_run(view, 9, exposure=1.0)
    assert o.stage_id == 2 and o.executions.tolist() == [2, 4, 3, 0, 0]
    rate1 = (o.stage_counts & 0xFF) + 1
    rate2 = ((o.stage_counts >> 8) & 0xFF) + 1
    assert rate1.min() >= 1 and rate1.max() <= MAX_RATE and rate2.min() >= 1 and rate2.max() <= MAX_RATE
    assert (o.stage_counts >> 16).max() == 0, "bytes of stages not built yet stay zero"
    assert rate2.max() > 1, "the scene must make the rates differ"
    # a pixel's sample count follows its block's rates
    n = o.pixel_samples().reshape(H, W)
    bx, by = np.arange(W) // 4, np.arange(H) // 4
    block = bx[None, :] + by[:, None] * o.blocks[0]
    assert np.array_equal(n, 2 + 4 * rate1[block] + 3 * rate2[block])
    # the beauty image is the mean over exactly those samples
    res = o.result(mode=0)
    want = o.fm.reshape(3, H, W) * (np.float32(1.0) / n.astype(np.float32))[None]
    assert np.array_equal(res, want)
    # sample distribution image: current rate / 256
    dist = o.result(mode=3)
    assert np.array_equal(dist[0], rate2[block].astype(np.float32) / np.float32(256.0))


def test_oracle_adaptive_equals_uniform_when_rates_are_one(tmp_path_factory):
    """With max rate 1 every execution of every stage is one sample per pixel: the sums equal a plain render of the same ids."""
    view = _scene(tmp_path_factory)
    tone = default_output_params(W, H, 1)
    o = oracle_lib.AdaptiveOracle(view, 1, 1, 1, exposure=0.0, tone=tone)
    o.render(5)
    assert o.stage_id == 2 and o.executions.tolist() == [1, 2, 2, 0, 0]
    fm, sm, _ = oracle_lib.render(view, 0, 5)
    assert np.array_equal(o.fm.reshape(3, -1), fm) and np.array_equal(o.sm, sm)


@pytest.mark.gpu
@pytest.mark.parametrize("exposure", [0.0, 1.5])
def test_adaptive_parity(tmp_path_factory, exposure):
    view = _scene(tmp_path_factory)
    executions = 2 + 4 + 8 + 3  # into stage 3
    o, tone = _oracle_run(view, executions, exposure)
    core = Core(0)
    try:
        core.upload(view)
        core.set_pixels(None)
        core.adaptive_begin(MAX_RATE, AVG_RATE, INTERVAL, exposure=exposure, tone=tone)
        core.adaptive_render(5)       # split on purpose: the schedule does not depend on how the calls are cut
        core.adaptive_render(executions - 5)
        info = core.adaptive_info()
        assert info["stage_id"] == o.stage_id == 3 and info["executions"] == o.executions.tolist()
        counts, variance = core.adaptive_download()
        assert np.array_equal(variance, o.block_variance), "block variances of the last stage build"
        assert np.float32(info["variance_total"]) == o.variance_total
        assert np.array_equal(counts, o.stage_counts), "per-block rates"
        fm, sm = core.accumulators()
        assert np.array_equal(fm, o.fm.reshape(3, -1)) and np.array_equal(sm, o.sm), "moments"
        for mode in range(4):
            got = core.generate_result(mode=mode, exposure=1.25, tone=tone)
            want = o.result(mode=mode, exposure=1.25, tone=tone)
            assert np.array_equal(got, want), "result image mode %d" % mode
        got = core.generate_result(mode=0, local_error_minimization=True, tone=tone)
        want = o.result(mode=0, local_error_minimization=True, tone=tone)
        assert np.array_equal(got, want), "local error minimisation"
        # output chain on the result image: the mean is already formed, so the sample count of the chain is 1
        p = default_output_params(W, H, 1)
        argb = core.generate_output(p, first_moment=core.generate_result(mode=0, tone=tone).reshape(3, -1))
        want_argb, _ = oracle_lib.generate_output(p, o.result(mode=0).reshape(3, -1))
        assert np.array_equal(argb, want_argb)
        core.adaptive_end()
    finally:
        core.close()


@pytest.mark.gpu
def test_result_image_without_adaptive_sampling(tmp_path_factory):
    """lumc_generate_result on a uniform render: beauty = first moment / n; local error minimisation equals the oracle."""
    view = _scene(tmp_path_factory)
    core = Core(0)
    try:
        core.upload(view)
        core.set_pixels(None)
        core.render(0, 6, samples_per_pass=3)
        fm, sm = core.accumulators()
        tone = default_output_params(W, H, 6)
        for lem in (False, True):
            got = core.generate_result(mode=0, local_error_minimization=lem, uniform_samples=6, tone=tone)
            want = oracle_lib.generate_result(W, H, fm, sm, 0, lem, 6, 1.0, tone)
            assert np.array_equal(got, want)
        for mode in (1, 2, 3):
            got = core.generate_result(mode=mode, uniform_samples=6, exposure=2.0, tone=tone)
            want = oracle_lib.generate_result(W, H, fm, sm, mode, False, 6, 2.0, tone)
            assert np.array_equal(got, want), mode
    finally:
        core.close()


@pytest.mark.gpu
def test_adaptive_rendering_through_the_api(tmp_path):
    """luminary_ext_render with the renderer settings' adaptive sampling: the requested output equals the oracle's adaptive run pushed
    through the oracle's result image and output chain; the settings' diagnostic output mode is honoured."""
    host = scenes.cornell_host(str(tmp_path), W, H, BOUNCES)
    s = host.get_settings()
    s.enable_adaptive_sampling = True
    s.adaptive_sampling_max_sampling_rate, s.adaptive_sampling_avg_sampling_rate, s.adaptive_sampling_update_interval = MAX_RATE, AVG_RATE, INTERVAL
    s.adaptive_sampling_exposure_aware = True
    host.set_settings(s)
    cam = host.get_camera()
    cam.exposure = 0.25
    cam.use_local_error_minimization = True
    host.set_camera(cam)
    view = oracle_lib.with_luts(host.device_scene())
    promise = host.request_output(7, W, H)   # 2 + 4 + 1 executions: one execution into stage 2
    host.render(9)
    handle = host.try_await_output(promise)
    assert handle is not None
    img, count, _ = host.get_image(handle)
    assert count == 7

    exposure = float(np.exp(np.float32(0.25)))
    tone = default_output_params(W, H, 1)
    tone.exposure = exposure
    o = oracle_lib.AdaptiveOracle(view, MAX_RATE, AVG_RATE, INTERVAL, exposure=exposure, tone=tone)
    o.render(7)
    want = oracle_lib.api_output(tone, o.result(mode=0, local_error_minimization=True, exposure=exposure, tone=tone))
    assert np.array_equal(img, want)
    fm, sm = host.accumulators()
    o.render(2)
    assert np.array_equal(fm, o.fm.reshape(3, -1)) and np.array_equal(sm, o.sm)
    host.release_output(handle)


@pytest.mark.gpu
@pytest.mark.parametrize("tonemap,max_rate,avg_rate", [(0, 256, 8), (1, 3, 9), (6, 2, 1)])
def test_adaptive_parity_other_settings(tmp_path_factory, tonemap, max_rate, avg_rate):
    """Tone curves of the compression factor (none, ACES, custom AgX), a rate cap of 256 with a high mean rate (large executions),
    a mean rate above the cap (clamped like adaptive_sampler_setup does)."""
    view = _scene(tmp_path_factory)
    tone = default_output_params(W, H, 1)
    tone.tonemap = tonemap
    tone.agx_slope, tone.agx_power, tone.agx_saturation = 1.1, 1.2, 0.9
    o = oracle_lib.AdaptiveOracle(view, max_rate, avg_rate, 1, exposure=2.0, tone=tone)
    o.render(1 + 2 + 2)
    core = Core(0)
    try:
        core.upload(view)
        core.set_pixels(None)
        core.adaptive_begin(max_rate, avg_rate, 1, exposure=2.0, tone=tone)
        core.adaptive_render(5)
        counts, variance = core.adaptive_download()
        assert np.array_equal(counts, o.stage_counts) and np.array_equal(variance, o.block_variance)
        fm, sm = core.accumulators()
        assert np.array_equal(fm, o.fm.reshape(3, -1)) and np.array_equal(sm, o.sm)
        assert np.array_equal(core.generate_result(mode=2, exposure=2.0, tone=tone), o.result(mode=2, exposure=2.0, tone=tone))
    finally:
        core.close()


def test_block_mask_partitions_the_blocks():
    from luminary_amd.distributed import block_mask, tile_pixels
    w, h, world = 100, 70, 3
    masks = [block_mask(w, h, r, world, tile=16) for r in range(world)]
    assert np.array_equal(sum(m.astype(int) for m in masks), np.ones(25 * 18, dtype=int)), "every block has exactly one owner"
    for r in range(world):  # a rank's blocks cover exactly its pixels
        px = tile_pixels(w, h, r, world, tile=16)
        owned = masks[r].reshape(18, 25)[(px // w) // 4, (px % w) // 4]
        assert owned.all() and int(masks[r].sum()) * 16 >= px.size


@pytest.mark.gpu
def test_partitioned_adaptive_rendering_equals_one_gpu(tmp_path_factory):
    """Two ranks emulated by two contexts on one GPU: own blocks only, block variances exchanged at every stage build (here: added in
    numpy, in the real thing one all-reduce). The summed frame, the rates and the variances equal the single-context run."""
    from luminary_amd.distributed import block_mask
    view = _scene(tmp_path_factory)
    tone = default_output_params(W, H, 1)
    executions = 2 + 4 + 5
    o, _ = _oracle_run(view, executions, 1.5)
    world = 2
    cores = [Core(0) for _ in range(world)]
    try:
        for r, c in enumerate(cores):
            c.upload(view)
            c.set_pixels(None)
            c.adaptive_begin(MAX_RATE, AVG_RATE, INTERVAL, exposure=1.5, tone=tone)
            c.adaptive_set_partition(block_mask(W, H, r, world, tile=8))
        remaining = executions
        while remaining > 0:
            before = sum(cores[0].adaptive_info()["executions"])
            for c in cores:
                c.adaptive_render(remaining)
            infos = [c.adaptive_info() for c in cores]
            assert infos[0]["executions"] == infos[1]["executions"] and infos[0]["build_pending"] == infos[1]["build_pending"]
            remaining -= sum(infos[0]["executions"]) - before
            if infos[0]["build_pending"]:
                parts = [c.adaptive_variance() for c in cores]
                assert not np.any((parts[0] != 0) & (parts[1] != 0)), "a block's variance comes from its owner only"
                full = parts[0] + parts[1]
                for c in cores:
                    c.adaptive_build_from(full)
        assert cores[0].adaptive_info()["stage_id"] == o.stage_id == 2
        for c in cores:
            counts, variance = c.adaptive_download()
            assert np.array_equal(counts, o.stage_counts) and np.array_equal(variance, o.block_variance)
        fm = sum(c.accumulators()[0] for c in cores)
        sm = sum(c.accumulators()[1] for c in cores)
        assert np.array_equal(fm, o.fm.reshape(3, -1)) and np.array_equal(sm, o.sm)
    finally:
        for c in cores:
            c.close()
