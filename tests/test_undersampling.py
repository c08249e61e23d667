"""Undersampling preview (second part) and supersampled output (f1): the frame's first sample rendered coarse to fine,
the coarse images shown meanwhile, the 2^ss box filter of the final image. HIP == oracle, bit for bit."""
import numpy as np
import pytest

import oracle_lib
from luminary_amd import scenes
from luminary_amd.core import Core, default_output_params, undersampling_schedule

W, H = 70, 38  # not multiples of the coarsest block


def _host(tmp_path, undersampling, supersampling=0, adaptive=False, width=W, height=H):
    host = scenes.cornell_host(str(tmp_path), width, height, 2)
    s = host.get_settings()
    s.undersampling, s.supersampling, s.enable_adaptive_sampling = undersampling, supersampling, adaptive
    s.adaptive_sampling_update_interval = 2
    host.set_settings(s)
    return host


@pytest.mark.gpu
def test_undersampled_first_sample_matches_the_oracle(tmp_path):
    host = _host(tmp_path, 3)
    view = oracle_lib.with_luts(host.device_scene())
    core = Core(0)
    try:
        core.upload(view)
        core.set_pixels(None)
        frame_fm = np.zeros((3, W * H), np.float32)
        frame_sm = np.zeros(W * H, np.float32)
        for stage, it in undersampling_schedule(3):
            core.render_undersampled(stage, it)
            px = oracle_lib.undersampling_pixels(W, H, stage, it)
            fm, sm, _ = oracle_lib.render(view, 0, 1, pixels=px)
            frame_fm[:, px] += fm
            frame_sm[px] += sm
            got_fm, got_sm = core.accumulators()
            assert np.array_equal(got_fm, frame_fm) and np.array_equal(got_sm, frame_sm), "stage %d iteration %d" % (stage, it)
            got = core.generate_result_undersampled(stage, it)
            want = oracle_lib.result_undersampled(frame_fm, W, H, stage, it)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "preview image of stage %d iteration %d" % (stage, it)
            # ... and through the display chain at the frame's size
            p = default_output_params(W, H, 1, undersampling_stage=stage)
            argb = core.generate_output(p, first_moment=got.reshape(3, -1))
            assert np.array_equal(argb, oracle_lib.generate_output(p, want.reshape(3, -1))[0])
        # the schedule is the first sample of every pixel: the same frame as one ordinary pass
        ofm, osm, _ = oracle_lib.render(view, 0, 1)
        assert np.array_equal(frame_fm, ofm) and np.array_equal(frame_sm, osm)
        core.clear()
        core.render(0, 1)
        plain_fm, plain_sm = core.accumulators()
        assert np.array_equal(plain_fm, frame_fm) and np.array_equal(plain_sm, frame_sm)
    finally:
        core.close()


@pytest.mark.gpu
@pytest.mark.parametrize("adaptive", [False, True])
def test_preview_through_the_host_api(tmp_path, adaptive):
    """luminary_ext_render with recurring outputs enabled: the first sample allocation is the preview schedule; the image on display after it
    is the stage-1 coarse image (device.c:1509-1536 produces the output before the state advances); later frames are the usual ones."""
    ss = 1
    host = _host(tmp_path, 2, supersampling=ss, adaptive=adaptive, width=48, height=32)
    view = oracle_lib.with_luts(host.device_scene())
    wi, hi = view.width, view.height
    assert (wi, hi) == (96, 64), "the frame is rendered at the output size << supersampling"
    host.set_output_properties(48, 32)
    at1 = host.request_output(1, 48, 32)
    host.render(1)
    ofm, osm, _ = oracle_lib.render(view, 0, 1)
    got_fm, got_sm = host.accumulators()
    assert np.array_equal(got_fm, ofm) and np.array_equal(got_sm, osm)
    coarse = oracle_lib.result_undersampled(ofm, wi, hi, 1, 0)
    p = default_output_params(wi, hi, 1, supersampling=ss, undersampling_stage=1)
    want_preview = oracle_lib.api_output(p, coarse)
    rec = host.acquire_output()
    img, count, _ = host.get_image(rec)
    assert count == 1 and np.array_equal(img, want_preview)
    h1 = host.try_await_output(at1)
    assert h1 is not None and np.array_equal(host.get_image(h1)[0], want_preview), "a request keyed to one sample gets the same image"
    host.release_output(rec)
    host.release_output(h1)
    # the second allocation is an ordinary sample; its image is the box-filtered full frame
    host.render(1)
    fm2, _, _ = oracle_lib.render(view, 0, 2)
    got_fm, _ = host.accumulators()
    assert np.array_equal(got_fm, fm2)
    q = default_output_params(wi, hi, 2, supersampling=ss)
    rec = host.acquire_output()
    img2, count2, _ = host.get_image(rec)
    assert count2 == 2 and np.array_equal(img2, oracle_lib.api_output(q, fm2 * (np.float32(1.0) / np.float32(2.0))))
    if adaptive:  # interval 2: stage 1 was built after these two executions, from the same moments as without a preview
        info = Core.adaptive_info_of(host.core_context())
        assert info["stage_id"] == 1 and info["executions"][0] == 2


@pytest.mark.gpu
def test_no_preview_without_recurring_outputs(tmp_path):
    """device_setup_undersampling (device.c:1302-1307): nobody watches, so the first sample is an ordinary pass; a request keyed to it gets
    the full image."""
    host = _host(tmp_path, 2, supersampling=1, width=48, height=32)
    view = oracle_lib.with_luts(host.device_scene())
    at1 = host.request_output(1, 48, 32)
    host.render(1)
    fm, _, _ = oracle_lib.render(view, 0, 1)
    q = default_output_params(view.width, view.height, 1, supersampling=1)
    h1 = host.try_await_output(at1)
    assert h1 is not None and np.array_equal(host.get_image(h1)[0], oracle_lib.api_output(q, fm))
