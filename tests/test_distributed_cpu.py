"""The N > 1 path on the CPU: world_size 2 over gloo. Each rank owns interleaved image tiles, renders them (with the oracle standing in
for the GPU renderer, which is allowed in tests), and one reduce assembles the frame on rank 0, as bench.py does over RCCL."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp, result_file, width=48, height=32, tile=8):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import oracle_lib
    from luminary_amd import scenes
    from luminary_amd.distributed import assemble_frame, gather_frame, tile_pixels
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    view = oracle_lib.with_luts(scenes.cornell_host(os.path.join(tmp, "r%d" % rank), width, height, 2).device_scene())
    px = tile_pixels(view.width, view.height, rank, world, tile=tile)
    fm, sm, _ = oracle_lib.render(view, 0, 2, pixels=px, threads=2)
    full = assemble_frame(torch.from_numpy(fm.reshape(-1)), torch.from_numpy(sm), px, view.width * view.height, dist, 0)
    # ... and the same frame by a gather of the ranks' own pixels (lumc_frame_gather's scheme: padded shares, the root derives every rank's pixel list)
    gathered = gather_frame(torch.from_numpy(fm.reshape(-1)), torch.from_numpy(sm), view.width, view.height, rank, world, dist, 0, tile=tile)
    if rank == 0:
        ref_fm, ref_sm, _ = oracle_lib.render(view, 0, 2, threads=2)
        ok = np.array_equal(full[:3].numpy(), ref_fm) and np.array_equal(full[3].numpy(), ref_sm)
        ok = ok and gathered is not None and np.array_equal(gathered.numpy(), full.numpy())
        open(result_file, "w").write("ok" if ok else "mismatch")
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_partition_and_reduce(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    result = os.path.join(str(tmp_path), "result.txt")
    mp.spawn(_worker, args=(2, port, str(tmp_path), result), nprocs=2, join=True)
    assert open(result).read() == "ok"


def test_four_rank_uneven_tiles_and_reduce(tmp_path):
    """World size 4 on a frame that is not a multiple of the tile (50 x 37 with 16-pixel tiles: 4 x 3 tiles, ragged right and bottom edges,
    12 tiles over 4 ranks; 1080 / 32 is not integral either): the reduce still assembles exactly the single-process frame."""
    import torch.multiprocessing as mp
    port = 33500 + (os.getpid() % 2000)
    result = os.path.join(str(tmp_path), "result4.txt")
    mp.spawn(_worker, args=(4, port, str(tmp_path), result, 50, 37, 16), nprocs=4, join=True)
    assert open(result).read() == "ok"


def _adaptive_worker(rank, world, port, tmp, result_file, w=40, h=24, tile=8):
    """Adaptive rendering partitioned over two ranks: luminary_amd.distributed.adaptive_render drives a per-rank renderer (the oracle
    standing in for the GPU core, same interface) and exchanges the block variances with one all-reduce per stage build."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import oracle_lib
    from luminary_amd import scenes
    from luminary_amd.core import default_output_params
    from luminary_amd.distributed import adaptive_render, block_mask
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    view = oracle_lib.with_luts(scenes.cornell_host(os.path.join(tmp, "a%d" % rank), w, h, 2).device_scene())
    tone = default_output_params(w, h, 1)
    core = oracle_lib.AdaptiveOracle(view, 5, 2, 1, exposure=1.0, tone=tone, threads=2)
    core.adaptive_set_partition(block_mask(w, h, rank, world, tile=tile))
    adaptive_render(core, 1 + 2 + 3, dist)
    frame = torch.from_numpy(np.concatenate([core.fm, core.sm]))
    dist.reduce(frame, dst=0, op=dist.ReduceOp.SUM)  # every pixel has one owner: the sum assembles the frame
    if rank == 0:
        ref = oracle_lib.AdaptiveOracle(view, 5, 2, 1, exposure=1.0, tone=tone, threads=2)
        ref.render(1 + 2 + 3)
        ok = (core.stage_id == ref.stage_id == 2 and np.array_equal(core.stage_counts, ref.stage_counts) and np.array_equal(core.block_variance, ref.block_variance)
              and np.array_equal(frame.numpy(), np.concatenate([ref.fm, ref.sm])))
        open(result_file, "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_adaptive_rendering(tmp_path):
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    result = os.path.join(str(tmp_path), "result_adaptive.txt")
    mp.spawn(_adaptive_worker, args=(2, port, str(tmp_path), result), nprocs=2, join=True)
    assert open(result).read() == "ok"


def test_four_rank_adaptive_rendering_uneven(tmp_path):
    """Adaptive rendering over four ranks on a frame whose size is neither a multiple of the tile nor of the 4 x 4 adaptive block (42 x 27):
    the one all-reduce of block variances per stage build gives every rank the rates of the single-process run."""
    import torch.multiprocessing as mp
    port = 35500 + (os.getpid() % 2000)
    result = os.path.join(str(tmp_path), "result_adaptive4.txt")
    mp.spawn(_adaptive_worker, args=(4, port, str(tmp_path), result, 42, 27, 8), nprocs=4, join=True)
    assert open(result).read() == "ok"


def _frame_worker(rank, world, port, result_file, width, height):
    """The partition and the one exchange step of BASELINE config 4 (4K frame over 8 ranks) without the rendering in between: every rank fills
    the accumulators of its own pixels with a function of the pixel index; the assembled frame must hold that function everywhere."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from luminary_amd.distributed import assemble_frame, tile_pixels
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    px = tile_pixels(width, height, rank, world)  # 32 x 32 tiles, round robin
    value = lambda p, c: ((p.astype(np.int64) * 2654435761 + c * 40503) % 1000003).astype(np.float32)
    fm = np.stack([value(px, c) for c in range(3)]).reshape(-1)
    sm = value(px, 3)
    counts = torch.tensor([px.size], dtype=torch.int64)
    dist.all_reduce(counts)
    full = assemble_frame(torch.from_numpy(fm), torch.from_numpy(sm), px, width * height, dist, 0)
    if rank == 0:
        p = np.arange(width * height, dtype=np.int64)
        ok = int(counts[0]) == width * height and all(np.array_equal(full[c].numpy(), value(p, c)) for c in range(4))
        open(result_file, "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_tile_deal_and_reduce_of_the_4k_frame(tmp_path):
    """BASELINE config 4's frame, 3840 x 2160 over 8 ranks: 120 x 67.5 tiles of 32 pixels (a ragged bottom row of 16-pixel-high tiles), 8 100 tiles
    dealt round robin, every pixel owned once, one reduce of 4 x 8.3 M floats per rank assembles the frame on rank 0."""
    import torch.multiprocessing as mp
    port = 37500 + (os.getpid() % 2000)
    result = os.path.join(str(tmp_path), "result_4k.txt")
    mp.spawn(_frame_worker, args=(8, port, result, 3840, 2160), nprocs=8, join=True)
    assert open(result).read() == "ok"
