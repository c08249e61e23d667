"""Multi-GPU behind the C ABI (include/lum_core.h lumc_tile_pixels / lumc_comm_* / lumc_frame_*, and the host API's device functions):
the frame is dealt to the GPUs in 32x32 tiles and assembled with one RCCL reduce per output. Reference: device_manager.c:776-862 (device
enumeration), device_result_interface.c:107-299 (its sample partition with host-staged sums, which this replaces).

What can run where: the tile deal and the API shape on the CPU; on the single-GPU test box the RCCL call path with a one-rank communicator,
and the host's partition / assembly logic with device 0 presented several times (LUM_FAKE_DEVICES; RCCL refuses two ranks on one GPU, so
that run assembles through the peer-copy transport). The N-rank RCCL reduce itself first runs in the driver's multi-GPU bench."""
import numpy as np
import pytest

import luminary_amd
import oracle_lib
from luminary_amd import scenes
from luminary_amd.core import Core
from luminary_amd.distributed import tile_pixels


def test_tile_deal_of_the_c_abi_matches_the_python_one_and_covers_the_frame():
    for (w, h, world) in ((1920, 1080, 8), (100, 70, 3), (33, 31, 4), (64, 64, 1), (50, 37, 4)):
        seen = np.zeros(w * h, dtype=np.int32)
        for rank in range(world):
            px = Core.tile_pixels(w, h, rank, world)
            assert np.array_equal(px, tile_pixels(w, h, rank, world)), (w, h, world, rank)
            seen[px] += 1
        assert (seen == 1).all(), "every pixel has exactly one owner"


def test_the_library_links_rccl():
    """The C library itself carries the RCCL reduce (north star: 'the C host ... tiles the image across the GPUs with an RCCL reduce')."""
    import subprocess
    out = subprocess.run(["ldd", luminary_amd.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" in out
    lib = luminary_amd._lib()
    for name in ("lumc_comm_unique_id", "lumc_comm_init_rank", "lumc_comm_init_all", "lumc_frame_assemble", "lumc_frame_assemble_all", "lumc_device_count"):
        assert hasattr(lib, name), name


def test_host_without_gpu_still_answers_device_queries():
    host = luminary_amd.Host()
    assert host.get_device_count() >= 1


@pytest.mark.gpu
def test_rccl_frame_assembly_with_a_one_rank_communicator():
    """lumc_comm_unique_id -> lumc_comm_init_rank -> lumc_frame_assemble: scatter + ncclReduce on the GPU box's one GPU. The assembled frame
    holds this rank's pixels at their frame positions and zeros elsewhere."""
    host = scenes.example_scene(160, 96, 3, sphere_segments=8, ground_res=12, num_objects=16, num_lights=4)
    view = oracle_lib.with_luts(host.device_scene())
    core = Core(0)
    try:
        core.upload(view)
        px = Core.tile_pixels(160, 96, 1, 3)
        core.set_pixels(px)
        core.render(0, 2, samples_per_pass=2)
        fm, sm = core.accumulators()
        core.comm_init_rank(1, 0, Core.comm_unique_id())
        assert core.frame_assemble(160 * 96, 0) != 0
        ffm, fsm = core.frame_download(160 * 96)
    finally:
        core.close()
    want = np.zeros((3, 160 * 96), dtype=np.float32)
    want[:, px] = fm
    want_sm = np.zeros(160 * 96, dtype=np.float32)
    want_sm[px] = sm
    assert np.array_equal(ffm, want) and np.array_equal(fsm, want_sm)
    ofm, osm, _ = oracle_lib.render(view, 0, 2, pixels=px)
    assert np.array_equal(fm, ofm)


@pytest.mark.gpu
def test_host_tiles_the_frame_over_its_devices(tmp_path, monkeypatch):
    """The host API with three device slots (device 0 three times on this box): luminary_host_get_device_count / _get_device_info report
    them, a whole-frame render is dealt to them in tiles, and accumulators, ray counters and the ARGB8 output equal the single-device
    render bit for bit; disabling devices (including the main one) re-elects the main device and restarts."""
    w, h = 100, 70  # not a multiple of the 32-pixel tile
    single = scenes.cornell_host(str(tmp_path / "one"), w, h, 3)
    single.set_output_properties(w, h)
    single.render(4)
    fm1, sm1 = single.accumulators()
    img1, n1, _ = single.get_image(single.acquire_output())
    cnt1 = single.ray_counters()

    monkeypatch.setenv("LUM_FAKE_DEVICES", "3")
    monkeypatch.setenv("LUM_MAX_DEVICES", "8")
    multi = scenes.cornell_host(str(tmp_path / "three"), w, h, 3)
    assert multi.get_device_count() == 3
    infos = [multi.get_device_info(i) for i in range(3)]
    assert [i.is_main_device for i in infos] == [True, False, False] and all(i.is_enabled for i in infos)
    multi.set_output_properties(w, h)
    multi.render(4)
    fm3, sm3 = multi.accumulators()
    img3, n3, _ = multi.get_image(multi.acquire_output())
    assert n1 == n3 == 4
    assert np.array_equal(fm3, fm1) and np.array_equal(sm3, sm1), "tiled accumulation == single device"
    assert np.array_equal(img3, img1), "same ARGB8 output"
    assert multi.ray_counters()[:4] == cnt1[:4], "ray counters add up over the devices"
    view = oracle_lib.with_luts(multi.device_scene())
    ofm, osm, _ = oracle_lib.render(view, 0, 4)
    assert np.array_equal(fm3, ofm) and np.array_equal(sm3, osm)

    multi.set_device_enable(0, False)  # the main device leaves: device 1 takes over, the integration restarts on two devices
    infos = [multi.get_device_info(i) for i in range(3)]
    assert [i.is_main_device for i in infos] == [False, True, False] and [i.is_enabled for i in infos] == [False, True, True]
    multi.render(2)
    fm2, _ = multi.accumulators()
    ofm2, _, _ = oracle_lib.render(view, 0, 2)
    assert np.array_equal(fm2, ofm2)
    multi.set_device_enable(2, False)
    multi.render(3)
    fmx, _ = multi.accumulators()
    ofm3, _, _ = oracle_lib.render(view, 0, 3)
    assert np.array_equal(fmx, ofm3), "one device left: the untiled path"
    with pytest.raises(luminary_amd.LuminaryError):
        multi.set_device_enable(1, False)  # the last device cannot be disabled


def test_bench_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` outside a torchrun environment starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 ...` as a child process before anything touches the GPU and returns its exit code."""
    import sys

    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "torch" not in sys.modules or not hasattr(sys.modules["torch"], "_lum_gpu_touched")


def test_bench_does_not_price_kernels_with_counters_of_another_build(monkeypatch):
    """The committed PMC record names the source tree (hash of the device code and build flags) and the LDS split of the library it was collected
    with; bench.py derives counter-based figures only when both match its own - otherwise the record is flagged stale and `traffic` / `frac` stay
    null (never an inferred bound)."""
    import bench
    SPP = 64  # bench.py's default pass (round 6; tools/pmc_collect.py collects at the same size)
    recorded = bench.pmc_record("hall", SPP, "fast", 0)
    if recorded is None:
        pytest.skip("no counter record for the hall in profiles/pmc_counters.json")
    then = int(recorded.get("lds_stack_bytes", 0))
    monkeypatch.setattr(bench, "source_hash", lambda: recorded.get("source_hash"))
    assert bench.pmc_record("hall", SPP, "fast", then)["_stale"] is False
    assert bench.pmc_record("hall", SPP, "fast", then + 4096)["_stale"] is True
    monkeypatch.setattr(bench, "source_hash", lambda: "another tree")
    assert bench.pmc_record("hall", SPP, "fast", then)["_stale"] is True
    assert bench.pmc_record("hall", SPP, "exact", then) is None or bench.pmc_record("hall", SPP, "exact", then).get("flavour") == "exact"


def _adaptive_rates(host):
    """stage counts and block variances of the host's main context (lumc_adaptive_download on Host.core_context())"""
    import ctypes as C
    lib = luminary_amd._lib()
    blocks = ((host.get_settings().width + 3) // 4) * ((host.get_settings().height + 3) // 4)
    counts = np.zeros(blocks, dtype=np.uint32)
    var = np.zeros(blocks, dtype=np.float32)
    assert lib.lumc_adaptive_download(C.c_void_p(host.core_context()), counts.ctypes.data_as(C.c_void_p), var.ctypes.data_as(C.c_void_p)) == 0
    return counts, var


def _configure(host, w, h, adaptive, undersampling):
    s = host.get_settings()
    s.undersampling, s.supersampling, s.enable_adaptive_sampling = undersampling, 0, adaptive
    s.adaptive_sampling_max_sampling_rate, s.adaptive_sampling_avg_sampling_rate, s.adaptive_sampling_update_interval = 8, 2, 2
    s.adaptive_sampling_exposure_aware = True
    host.set_settings(s)
    host.set_output_properties(w, h)


@pytest.mark.gpu
@pytest.mark.parametrize("adaptive,undersampling", [(True, 0), (False, 2), (True, 2)])
def test_host_tiles_adaptive_sampling_and_keeps_the_preview(tmp_path, monkeypatch, adaptive, undersampling):
    """Default-settings frontends on several GPUs (the reference: device_adaptive_sampler.c:205-213, device_manager.c:452-469 - adaptive sampling is
    its default mode): with three device slots the host tiles adaptive rendering too (every device owns the blocks of its tiles, one exchange of
    block variances per stage) and renders the frame's first sample as the undersampling preview on the main device, whose sums the tiles then take
    over. Accumulators, rates, block variances, the recurring ARGB8 output and the ray counters equal the single-device run bit for bit."""
    w, h = 100, 70
    frames = {}
    for name, fake in (("one", None), ("three", "3")):
        if fake:
            monkeypatch.setenv("LUM_FAKE_DEVICES", fake)
            monkeypatch.setenv("LUM_MAX_DEVICES", "8")
        host = scenes.cornell_host(str(tmp_path / name), w, h, 3)
        assert host.get_device_count() == (3 if fake else 1)
        _configure(host, w, h, adaptive, undersampling)
        host.render(1)                      # the preview (if any) alone: its coarse images are what is on display
        img_first, n_first, _ = host.get_image(host.acquire_output())
        host.render(9)                      # through two stage builds at update interval 2 (2 + 4 executions)
        fm, sm = host.accumulators()
        img, n, _ = host.get_image(host.acquire_output())
        rates = _adaptive_rates(host) if adaptive else None
        frames[name] = (fm, sm, img, n, img_first, n_first, rates, host.ray_counters()[:4])
        host.close()
    a, b = frames["one"], frames["three"]
    assert a[3] == b[3] == 10 and a[5] == b[5] == 1
    assert np.array_equal(a[4], b[4]), "the image after the first allocation (preview)"
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), "accumulators"
    assert np.array_equal(a[2], b[2]), "ARGB8 output"
    assert a[7] == b[7], "ray counters add up over the devices"
    if adaptive:
        assert np.array_equal(a[6][0], b[6][0]), "per-block rates"
        assert np.array_equal(a[6][1], b[6][1]), "block variances of the last build"
        assert len(np.unique(a[6][0])) > 1, "the rates differ over the frame (otherwise the test proves little)"


@pytest.mark.gpu
def test_bench_under_torchrun_reduces_through_the_c_abi():
    """The command line the driver's multi-GPU bench uses, with one rank (what a one-GPU box offers): `python -m torch.distributed.run --nproc-per-node 1
    bench.py --gpus 1 ...` must take the distributed path - tile deal, the library's own RCCL communicator made from rank 0's id, lumc_frame_assemble -
    and say so in its line; a communicator failure ends the run non-zero (bench.py), so rc 0 here means RCCL took the frame."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("LUM_FLAVOUR", None)  # the bench's own default
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "1", "--workload", "cornell", "--width", "256", "--height", "192", "--steps", "1", "--warmup", "1",
           "--samples-per-pass", "4", "--cpu-budget", "0", "--secondary", "none"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    assert len(line) <= 4096
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0
    assert out["config"]["frame_reduce"].startswith("C ABI: lumc_frame_gather"), out["config"]["frame_reduce"]
    assert out["config"]["rccl_ranks"] == 1
    assert out["config"]["partition"].startswith("32x32")


@pytest.mark.gpu
def test_rccl_tile_gather_with_a_one_rank_communicator():
    """lumc_frame_gather on the box's one GPU: the context holds share 0 of 1 of the tile deal (the whole frame, in tile order), packs it, ncclGather with a
    one-rank communicator, the scatter by the deal's pixel list - the frame must hold every pixel's sums at its place. A pixel set that is not a share of the
    deal is refused (those reduce)."""
    w, h = 100, 70
    host = scenes.example_scene(w, h, 3, sphere_segments=8, ground_res=12, num_objects=16, num_lights=4)
    view = oracle_lib.with_luts(host.device_scene())
    core = Core(0)
    try:
        core.upload(view)
        px = Core.tile_pixels(w, h, 0, 1)
        core.set_pixels(px)
        core.render(0, 2, samples_per_pass=2)
        fm, sm = core.accumulators()
        core.comm_init_rank(1, 0, Core.comm_unique_id())
        assert core.frame_gather(w, h, 0)
        ffm, fsm = core.frame_download(w * h)
        core.set_pixels(px[: px.size // 2])
        from luminary_amd.core import CoreError
        with pytest.raises(CoreError):
            core.frame_gather(w, h, 0)
        core.set_pixels(None)  # every pixel, but in row-major order: not the deal's order either (the root would scatter the sums to the wrong pixels)
        with pytest.raises(CoreError):
            core.frame_gather(w, h, 0)
        core.set_pixels(px[::-1].copy())  # the right pixels in another order
        with pytest.raises(CoreError):
            core.frame_gather(w, h, 0)
    finally:
        core.close()
    want = np.zeros((3, w * h), dtype=np.float32)
    want[:, px] = fm
    want_sm = np.zeros(w * h, dtype=np.float32)
    want_sm[px] = sm
    assert np.array_equal(ffm, want) and np.array_equal(fsm, want_sm)
    ofm, osm, _ = oracle_lib.render(view, 0, 2)
    assert np.array_equal(ffm, ofm) and np.array_equal(fsm, osm)


@pytest.mark.gpu
def test_tile_gather_over_three_contexts_equals_the_reduce_and_the_single_render():
    """lumc_frame_gather_all with three contexts (all on device 0: RCCL refuses that, so the packed buffers travel by peer copy - the pack, the padding of
    uneven shares and the root's scatter through the other ranks' pixel lists are what is under test) against lumc_frame_assemble_all and a full-frame render."""
    import ctypes as C
    w, h = 100, 70  # 4 x 3 tiles of 32: shares of 4 / 4 / 4 tiles with ragged right and bottom edges
    host = scenes.example_scene(w, h, 3, sphere_segments=8, ground_res=12, num_objects=16, num_lights=4)
    view = oracle_lib.with_luts(host.device_scene())
    cores = [Core(0) for _ in range(3)]
    try:
        for r, c in enumerate(cores):
            c.upload(view)
            c.set_pixels(Core.tile_pixels(w, h, r, 3))
            c.render(0, 2, samples_per_pass=2)
        lib = luminary_amd._lib()
        arr = (C.c_void_p * 3)(*[c._ctx for c in cores])
        out = C.c_void_p()
        assert lib.lumc_frame_gather_all(arr, 3, C.c_uint32(w), C.c_uint32(h), 0, C.byref(out)) == 0
        gfm, gsm = cores[0].frame_download(w * h)
        assert lib.lumc_frame_assemble_all(arr, 3, C.c_uint32(w * h), 0, C.byref(out)) == 0
        rfm, rsm = cores[0].frame_download(w * h)
    finally:
        for c in cores:
            c.close()
    assert np.array_equal(gfm, rfm) and np.array_equal(gsm, rsm), "gather == reduce"
    ofm, osm, _ = oracle_lib.render(view, 0, 2)
    assert np.array_equal(gfm, ofm) and np.array_equal(gsm, osm), "... == the frame rendered in one piece"


def test_the_deal_is_a_lattice_and_not_column_stripes():
    """Round 5: tile (x, y) -> rank (x + k*y) % world. Rounds 1-4 dealt t % world over the row-major grid, which at 3840 px (120 tiles per row, 120 % 8 == 0)
    gave every rank vertical 32-pixel stripes. Under the lattice every row AND every column of tiles of C4's frame holds all 8 ranks, a rank's nearest own
    tiles are >= 2.8 tiles away in every direction, the shares are equal to 0.5 %, and the step does not depend on the frame."""
    from luminary_amd.distributed import tile_lattice_step, tile_owner, tile_share_counts
    lib = luminary_amd._lib()
    assert [tile_lattice_step(n) for n in (1, 2, 4, 6, 8)] == [0, 1, 1, 1, 3], "steps coprime to the rank count (round 6: k = 2 at 4 ranks sent every row's spare tiles to the same two ranks)"
    import math
    assert all(math.gcd(tile_lattice_step(n), n) == 1 for n in range(2, 40))
    assert int(lib.lumc_tile_lattice_step(72)) == tile_lattice_step(72)  # beyond the precomputed table: cached per size, not recomputed per tile
    for (w, h, n) in ((1376, 1080, 4), (1366, 768, 4), (1366, 768, 6)):  # widths that are no multiple of world tiles (the advisor's cases: 1.023 / 1.016 with k = 2)
        counts = tile_share_counts(w, h, n)
        assert max(counts) / (w * h / n) < 1.012, (w, h, n, counts)
    assert [int(lib.lumc_tile_lattice_step(n)) for n in range(1, 17)] == [tile_lattice_step(n) for n in range(1, 17)]
    for (w, h) in ((3840, 2160), (1920, 1080)):
        tx, ty = (w + 31) // 32, (h + 31) // 32
        i, j = np.meshgrid(np.arange(tx), np.arange(ty), indexing="xy")
        owner = tile_owner(i, j, tx, 8)
        for row in owner:
            for x0 in range(0, tx - 7):
                assert len(set(row[x0:x0 + 8].tolist())) == 8
        for col in owner.T:
            for y0 in range(0, ty - 7):
                assert len(set(col[y0:y0 + 8].tolist())) == 8
        ys, xs = np.nonzero(owner == 5)
        d2 = (xs[:, None] - xs[None]) ** 2 + (ys[:, None] - ys[None]) ** 2
        assert d2[d2 > 0].min() == 8, "nearest tile of the same rank at (2, 2)"
        counts = tile_share_counts(w, h, 8)
        assert sum(counts) == w * h and max(counts) / (w * h / 8) < 1.005
        assert counts == [Core.tile_pixels(w, h, r, 8).size for r in range(8)]


def _eight_slot_host_equals_one_device(make_host, tmp_path, monkeypatch, spp):
    frames = {}
    for name, fake in (("one", None), ("eight", "8")):
        if fake:
            monkeypatch.setenv("LUM_FAKE_DEVICES", fake)
            monkeypatch.setenv("LUM_MAX_DEVICES", "8")
        host = make_host()
        assert host.get_device_count() == (8 if fake else 1)
        s = host.get_settings()
        host.set_output_properties(s.width, s.height)
        host.render(spp)
        fm, sm = host.accumulators()
        img, n, _ = host.get_image(host.acquire_output())
        frames[name] = (fm, sm, img, n, host.ray_counters()[:4])
        host.close()
    a, b = frames["one"], frames["eight"]
    assert a[3] == b[3] == spp
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), "accumulators of 8 tile shares == one device"
    assert np.array_equal(a[2], b[2]), "ARGB8 output"
    assert a[4] == b[4], "ray counters add up over the 8 devices"
    assert float(a[0].max()) > 0.0


@pytest.mark.gpu
def test_c4_frame_over_eight_device_slots(tmp_path, monkeypatch):
    """BASELINE config 4's frame (Example-class scene, 3840x2160, 8 bounces) through the host API with EIGHT device slots (device 0 eight times: the host's
    partition, lumc_frame_gather_all over 8 shares by the peer-copy transport - RCCL refuses two ranks on one GPU -, the root's scatter through eight pixel
    lists of the lattice deal): accumulators, ARGB8 and ray counters equal the one-device render bit for bit. The N-rank RCCL gather itself first runs in the
    driver's multi-GPU bench."""
    _eight_slot_host_equals_one_device(lambda: scenes.example_scene(3840, 2160, 8), tmp_path, monkeypatch, 2)


@pytest.mark.gpu
def test_c5_scan_over_eight_device_slots(tmp_path, monkeypatch):
    """BASELINE config 5's scene (10 M-triangle scan, 1920x1080, 8 bounces) with eight device slots: eight contexts share the mesh's tree (LUM_BVH_SHARE), every
    one holds its own replica of the scene arrays (8 x 1.1 GB of 288 GB)."""
    _eight_slot_host_equals_one_device(lambda: scenes.scan_scene(1920, 1080, 8, triangles=10_000_000), tmp_path, monkeypatch, 2)


@pytest.mark.gpu
def test_an_hdri_bake_of_another_size_leaves_the_gather_buffers_alone():
    """lumc_sky_hdri_build freed the tile gather's three device buffers when the panorama's size changed and kept their addresses -
    the next lumc_frame_gather packed into freed memory and lumc_context_destroy freed them a second time. Gather, bake at a new size, gather, destroy."""
    import test_sky
    host = test_sky._scene(altitude=0.3)
    view = test_sky._with_sky_luts(host.device_scene())
    w, h = view.width, view.height
    core = Core(0)
    try:
        core.upload(view)
        px = Core.tile_pixels(w, h, 0, 1)
        core.set_pixels(px)
        core.render(0, 2, samples_per_pass=2)
        core.comm_init_rank(1, 0, Core.comm_unique_id())
        assert core.frame_gather(w, h, 0)
        first = core.frame_download(w * h)
        core.sky_hdri_build((1.0, 6.0, 28.0), 16, 4)
        core.sky_hdri_build((1.0, 6.0, 28.0), 40, 4)   # another size: the old panorama is freed - and only the panorama
        filler = [Core(0) for _ in range(2)]             # allocations that would land in freed blocks
        for f in filler:
            f.upload(view)
        assert core.frame_gather(w, h, 0)
        second = core.frame_download(w * h)
        for f in filler:
            f.close()
    finally:
        core.close()
    assert np.array_equal(first[0], second[0]) and np.array_equal(first[1], second[1])


def test_the_deal_on_random_frames_and_rank_counts():
    """Property test of the lattice deal (C library and Python twin): for random frame sizes, tile sizes and rank counts every pixel has exactly one owner, the
    two implementations agree rank by rank, the share sizes sum to the frame, and the ranks' tile counts differ by at most tiles_y % world."""
    from hypothesis import given, settings, strategies as st
    from luminary_amd.distributed import tile_owner, tile_share_counts

    @settings(max_examples=60, deadline=None)
    @given(w=st.integers(1, 700), h=st.integers(1, 500), world=st.integers(1, 16), tile=st.sampled_from([4, 8, 16, 32, 64]))
    def check(w, h, world, tile):
        seen = np.zeros(w * h, dtype=np.int32)
        counts = []
        for rank in range(world):
            px = Core.tile_pixels(w, h, rank, world, tile)
            assert np.array_equal(px, tile_pixels(w, h, rank, world, tile))
            seen[px] += 1
            counts.append(int(px.size))
        assert (seen == 1).all()
        assert counts == tile_share_counts(w, h, world, tile)
        tx, ty = (w + tile - 1) // tile, (h + tile - 1) // tile
        if tx >= world and ty >= world:  # frames of at least world x world tiles: the lattice gives every rank every row and column
            i, j = np.meshgrid(np.arange(tx), np.arange(ty), indexing="xy")
            owner = tile_owner(i, j, tx, world)
            per_rank = np.bincount(owner.ravel(), minlength=world)
            assert per_rank.max() - per_rank.min() <= ty % world, "over any `world` consecutive tile rows the ranks own the same number of tiles (the step is coprime to world)"

    check()
