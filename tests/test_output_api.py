"""Output side of the public API: promises and handles (host.h:72-87, :123), PNG files. CPU part: the store's rules and the PNG
writer; GPU part: images delivered through the API equal the oracle's output chain applied to the oracle's render."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

import luminary_amd
import oracle_lib
from luminary_amd import LuminaryError, scenes
from luminary_amd.core import default_output_params

INVALID_ARG, API_EXCEPTION = 3, 7


def _decode_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(typ + body), typ
        chunks.append((typ, body))
        pos += 12 + n
    assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, colour, comp, filt, interlace = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, colour, comp, filt, interlace) == (8, 6, 0, 0, 0)
    raw = zlib.decompress(chunks[1][1])
    rows = np.frombuffer(raw, dtype=np.uint8).reshape(h, 1 + 4 * w)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, 4)


def test_png_writer_round_trip(tmp_path):
    lib = luminary_amd._lib()
    lib.luminary_ext_write_png.restype = C.c_uint64
    rng = np.random.RandomState(0)
    w, h, ld = 37, 21, 40
    img = rng.randint(0, 2 ** 32, size=(h, ld), dtype=np.uint64).astype(np.uint32)
    path = str(tmp_path / "a.png")
    assert lib.luminary_ext_write_png(path.encode(), img.ctypes.data_as(C.c_void_p), C.c_uint32(w), C.c_uint32(h), C.c_size_t(ld)) == 0
    rgba = _decode_png(path)
    words = img[:, :w]
    want = np.stack([(words >> 16) & 0xFF, (words >> 8) & 0xFF, words & 0xFF, words >> 24], axis=-1).astype(np.uint8)
    assert np.array_equal(rgba, want)
    assert lib.luminary_ext_write_png(path.encode(), img.ctypes.data_as(C.c_void_p), C.c_uint32(0), C.c_uint32(h), C.c_size_t(ld)) == INVALID_ARG


def test_output_handles_without_rendering(tmp_path):
    host = scenes.cornell_host(str(tmp_path), 32, 32, 1)
    host.set_output_properties(32, 32)
    p0 = host.request_output(4, 32, 32)
    p1 = host.request_output(0, 64, 48)
    assert p0 != p1
    assert host.try_await_output(p0) is None and host.try_await_output(p1) is None and host.acquire_output() is None
    with pytest.raises(LuminaryError) as e:
        host.request_output(1, 1, 32)  # images are at least 2 x 2
    assert e.value.code == INVALID_ARG
    with pytest.raises(LuminaryError) as e:
        host.release_output(5)  # never handed out
    assert e.value.code == API_EXCEPTION
    host.release_output(luminary_amd.OUTPUT_HANDLE_INVALID)  # releasing "no output" is allowed (host_output_handler.c:150-152)
    with pytest.raises(LuminaryError) as e:
        host.save_png(luminary_amd.OUTPUT_HANDLE_INVALID, str(tmp_path / "x.png"))
    assert e.value.code == INVALID_ARG


@pytest.mark.gpu
def test_outputs_through_the_api_match_the_oracle(tmp_path):
    w, h = 64, 48
    host = scenes.cornell_host(str(tmp_path), w, h, 3)
    cam = host.get_camera()
    cam.exposure = 0.5
    host.set_camera(cam)
    host.set_output_properties(w, h)
    at2 = host.request_output(2, w, h)          # exactly at 2 samples
    scaled = host.request_output(0, 100, 30)    # the next output, other size
    at3 = host.request_output(3, w, h)          # the render loop stops at every requested count inside a render call
    host.render_samples(0, 4, samples_per_pass=4)

    view = oracle_lib.with_luts(host.device_scene())

    def oracle_image(spp, dst=None):
        fm, _, _ = oracle_lib.render(view, 0, spp)
        p = default_output_params(w, h, spp, dst=dst)
        p.exposure = float(np.exp(np.float32(0.5)))
        return oracle_lib.api_output(p, fm * (np.float32(1.0) / np.float32(spp)))  # result image, bloom, display chain

    h2 = host.try_await_output(at2)
    assert h2 is not None
    img, count, _ = host.get_image(h2)
    assert count == 2 and np.array_equal(img, oracle_image(2))
    hs = host.try_await_output(scaled)
    img_s, count_s, _ = host.get_image(hs)
    assert img_s.shape == (30, 100) and count_s == 2 and np.array_equal(img_s, oracle_image(2, dst=(100, 30)))
    h3 = host.try_await_output(at3)
    assert h3 is not None and host.get_image(h3)[1] == 3 and np.array_equal(host.get_image(h3)[0], oracle_image(3))
    rec = host.acquire_output()
    img_r, count_r, t = host.get_image(rec)
    ms = C.c_double()
    luminary_amd._call("luminary_host_get_current_sample_time", host._h, C.byref(ms))
    assert 0.0 < ms.value < 1e4, "milliseconds the latest sample took (device_sampletime.c)"
    assert count_r == 4 and t > 0.0 and np.array_equal(img_r, oracle_image(4))
    assert len({h2, hs, h3, rec}) == 4  # images owed to promises are not recycled before they were awaited

    path = str(tmp_path / "frame.png")
    host.save_png(rec, path)
    rgba = _decode_png(path)
    assert np.array_equal(rgba[..., 0], (img_r >> 16) & 0xFF) and np.array_equal(rgba[..., 2], img_r & 0xFF) and (rgba[..., 3] == 255).all()
    for handle in (h2, hs, h3, rec):
        host.release_output(handle)
    with pytest.raises(LuminaryError):
        host.release_output(rec)  # already released


@pytest.mark.gpu
def test_pixel_query_reports_the_first_hit(tmp_path):
    """luminary_host_get_pixel_info (host.h:89): instance, material, depth and the hit offset of the pixel's first-sample camera ray,
    checked against the oracle's camera ray and closest-hit query."""
    w, h = 64, 48
    host = scenes.cornell_host(str(tmp_path), w, h, 1)
    view = oracle_lib.with_luts(host.device_scene())
    l = oracle_lib.lib()
    import ctypes as C
    hits = 0
    for (x, y) in [(5, 5), (32, 24), (50, 40), (63, 47), (20, 10)]:
        r = host.get_pixel_info(x, y)
        ray = (C.c_float * 6)()
        l.oracle_camera_ray(C.byref(view), C.c_uint32(x), C.c_uint32(y), C.c_uint32(0), ray)
        o = np.array([list(ray)[:3]], dtype=np.float32)
        d = np.array([list(ray)[3:]], dtype=np.float32)
        want = oracle_lib.trace_closest(view, o, d, np.full((1, 2), 0xFFFFFFFF, dtype=np.uint32), use_bvh=False)[0]
        depth = want[2:3].copy().view(np.float32)[0]
        assert r.pixel_query_is_valid and np.float32(r.depth) == depth
        if want[0] < 0x7FFFFFFF:
            hits += 1
            assert r.instance_id == want[0]
            rel = (d[0] * depth).astype(np.float32).view(np.uint32) & 0xFFFF0000
            got = np.array([r.rel_hit_pos.x, r.rel_hit_pos.y, r.rel_hit_pos.z], dtype=np.float32).view(np.uint32)
            assert np.array_equal(got, rel) and r.material_id != 0xFFFF
    assert hits >= 3
    assert not host.get_pixel_info(w, h).pixel_query_is_valid  # outside the frame


def _build_cli(tmp_path):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    luminary_amd._lib()  # makes sure the library is built
    exe = str(tmp_path / "luminary_cli")
    lib_dir = os.path.join(root, "luminary_amd", "lib")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "luminary_cli.c"), "-L", lib_dir,
                           "-lluminary_amd", "-Wl,-rpath," + lib_dir, "-o", exe])
    return exe


def test_c_frontend_builds_and_fails_loudly_without_a_gpu(tmp_path):
    """examples/luminary_cli.c uses the public header only. Without a GPU the render call reports an error (no CPU fallback exists)."""
    import os
    import subprocess
    exe = _build_cli(tmp_path)
    scenes.cornell_box_files(str(tmp_path), 32, 24, 2)
    r = subprocess.run([exe, str(tmp_path / "cornell.lum"), "2", str(tmp_path / "out.png")], capture_output=True, text=True)
    if os.path.exists("/dev/kfd"):  # this test also runs on the GPU box: there the program simply works
        assert r.returncode == 0 and (tmp_path / "out.png").exists(), r.stderr
    else:
        assert r.returncode == 2 and "luminary_ext_render" in r.stderr and not (tmp_path / "out.png").exists(), (r.returncode, r.stderr)


@pytest.mark.gpu
def test_c_frontend_renders_the_same_png_as_the_python_binding(tmp_path):
    """A .lum file with the reference's default settings (adaptive sampling on, supersampling 1) through a C program that only knows
    include/luminary_amd.h, and the same calls through the ctypes mirror: identical PNG bytes."""
    import subprocess
    exe = _build_cli(tmp_path)
    scenes.cornell_box_files(str(tmp_path), 48, 32, 3)
    lum = str(tmp_path / "cornell.lum")
    r = subprocess.run([exe, lum, "5", str(tmp_path / "c.png")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "5 samples" in r.stdout
    host = luminary_amd.Host()
    host.load_lum_file(lum)
    s = host.get_settings()
    s.undersampling = 0
    host.set_settings(s)
    promise = host.request_output(5, s.width, s.height)
    host.render(5)
    handle = host.try_await_output(promise)
    host.save_png(handle, str(tmp_path / "py.png"))
    assert open(str(tmp_path / "c.png"), "rb").read() == open(str(tmp_path / "py.png"), "rb").read()
    assert _decode_png(str(tmp_path / "c.png")).shape == (s.height, s.width, 4)


def _build_unchanged_frontend(tmp_path):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    luminary_amd._lib()
    exe = str(tmp_path / "lum_bench")
    lib_dir = os.path.join(root, "luminary_amd", "lib")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "luminary_bench_unchanged.c"), "-L",
                           lib_dir, "-lluminary_amd", "-Wl,-rpath," + lib_dir, "-o", exe])
    return exe


def test_unchanged_frontend_compiles_against_the_forwarding_headers(tmp_path):
    """examples/luminary_bench_unchanged.c is Mandarin Duck's benchmark loop written against <luminary/*.h> and the reference's functions
    only (no luminary_ext_*): include/luminary/ forwards every header of the reference's include/luminary/ to luminary_amd.h."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("luminary", "host", "structs", "error", "path", "api_utils", "array", "queue", "ringbuffer", "host_memory", "log", "thread_status", "name_strings"):
        assert os.path.exists(os.path.join(root, "include", "luminary", name + ".h")), name
    exe = _build_unchanged_frontend(tmp_path)
    assert os.path.exists(exe)
    src = open(os.path.join(root, "examples", "luminary_bench_unchanged.c")).read()
    assert "luminary_ext_" not in src.split("*/", 1)[1]


def test_queue_workers_are_named_like_the_reference(tmp_path):
    host = luminary_amd.Host()
    workers = host.queue_workers()
    assert [w[0] for w in workers] == ["Host", "Device"] and all(w[1] is None for w in workers)


@pytest.mark.gpu
def test_unchanged_frontend_renders_on_the_library_thread(tmp_path):
    """The benchmark loop of an unchanged frontend: request outputs at 1, 2, 3, 4, 6, 8 samples, luminary_host_start_new_render, poll
    luminary_host_try_await_output. The library's Device worker renders; every requested image arrives, with the bytes the synchronous
    loop produces for the same sample count (default .lum settings: adaptive sampling on, supersampling 1)."""
    import subprocess
    exe = _build_unchanged_frontend(tmp_path)
    scenes.cornell_box_files(str(tmp_path), 48, 32, 3)
    lum = str(tmp_path / "cornell.lum")
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([exe, lum, "3", "cornell", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [l.split(",") for l in open(str(out / "BenchResults-cornell.txt")).read().strip().split("\n")]
    assert sorted(int(l[0]) for l in lines) == [1, 2, 3, 4, 6, 8]
    assert "queue worker 1: Device" in r.stdout and "queue worker 0: Host" in r.stdout
    host = luminary_amd.Host()
    host.load_lum_file(lum)
    s = host.get_settings()
    host.set_output_properties(0, 0, enabled=False)
    for count, more in ((3, 3), (8, 5)):
        promise = host.request_output(count, s.width, s.height)
        host.render(more)
        handle = host.try_await_output(promise)
        host.save_png(handle, str(tmp_path / ("py%d.png" % count)))
        assert open(str(out / ("Bench-%05d-cornell.png" % count)), "rb").read() == open(str(tmp_path / ("py%d.png" % count)), "rb").read(), count


@pytest.mark.gpu
def test_render_thread_restarts_on_edits_and_stops(tmp_path):
    """luminary_host_start_new_render starts the Device worker; images keep arriving through luminary_host_acquire_output; a camera move
    restarts the accumulation by itself; luminary_ext_stop_render hands control back to the synchronous calls."""
    import time
    w, h = 48, 32
    host = scenes.cornell_host(str(tmp_path), w, h, 2)
    host.set_output_properties(w, h)
    host.start_new_render()
    deadline = time.time() + 60
    count = 0
    while count < 12 and time.time() < deadline:
        handle = host.acquire_output()
        if handle is not None:
            _, count, _ = host.get_image(handle)
            host.release_output(handle)
        time.sleep(0.002)
    assert count >= 12, "the render thread keeps accumulating"
    assert host.is_rendering()[0]
    cam = host.get_camera()
    cam.pos.x += 0.05
    host.set_camera(cam)  # dirties the integration: the worker starts over on its own
    seen_restart = False
    deadline = time.time() + 60
    while time.time() < deadline and not seen_restart:
        handle = host.acquire_output()
        if handle is not None:
            _, c2, _ = host.get_image(handle)
            host.release_output(handle)
            seen_restart = c2 < count
        time.sleep(0.001)
    assert seen_restart, "a camera move restarts the accumulation"
    host.stop_render()
    running, n = host.is_rendering()
    assert not running
    time.sleep(0.05)
    assert host.is_rendering()[1] == n, "nothing renders after luminary_ext_stop_render"
    # the accumulated frame is the deterministic one: n samples of the moved camera
    view = oracle_lib.with_luts(host.device_scene())
    fm, _ = host.accumulators()
    ofm, _, _ = oracle_lib.render(view, 0, n)
    assert np.array_equal(fm, ofm), "asynchronously rendered frame == oracle at %d samples" % n


@pytest.mark.gpu
def test_output_only_changes_keep_the_accumulated_frame(tmp_path):
    """camera.c:80-147: exposure, tone curve, filter, bloom ... only change how the frame is shown (SCENE_DIRTY_FLAG_OUTPUT); moving the
    camera restarts the integration."""
    w, h = 48, 32
    host = scenes.cornell_host(str(tmp_path), w, h, 2)
    host.set_output_properties(w, h)
    view = oracle_lib.with_luts(host.device_scene())
    host.render(2)
    cam = host.get_camera()
    cam.exposure, cam.tonemap, cam.bloom_blend = 1.0, 1, 0.0
    host.set_camera(cam)
    host.render(1)
    fm, _ = host.accumulators()
    ofm, _, _ = oracle_lib.render(view, 0, 3)
    assert np.array_equal(fm, ofm), "three samples accumulated across the change"
    img, count, _ = host.get_image(host.acquire_output())
    p = default_output_params(w, h, 3)
    p.exposure, p.tonemap = float(np.exp(np.float32(1.0))), 1
    assert count == 3 and np.array_equal(img, oracle_lib.api_output(p, ofm * (np.float32(1.0) / np.float32(3.0)), blend=0.0)), "shown with the new exposure and curve"
    cam.pos.x += 0.1
    host.set_camera(cam)
    host.render(1)
    img, count, _ = host.get_image(host.acquire_output())
    assert count == 1, "a camera move starts over"
