"""The N > 1 path on the CPU: world_size 2 over gloo. Each rank owns interleaved image tiles, renders them (with the oracle standing in
for the GPU renderer, which is allowed in tests), and one reduce assembles the frame on rank 0, as bench.py does over RCCL."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp, result_file):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import oracle_lib
    from luminary_amd import scenes
    from luminary_amd.distributed import assemble_frame, tile_pixels
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    view = oracle_lib.with_luts(scenes.cornell_host(os.path.join(tmp, "r%d" % rank), 48, 32, 2).device_scene())
    px = tile_pixels(view.width, view.height, rank, world, tile=8)
    fm, sm, _ = oracle_lib.render(view, 0, 2, pixels=px, threads=2)
    full = assemble_frame(torch.from_numpy(fm.reshape(-1)), torch.from_numpy(sm), px, view.width * view.height, dist, 0)
    if rank == 0:
        ref_fm, ref_sm, _ = oracle_lib.render(view, 0, 2, threads=2)
        ok = np.array_equal(full[:3].numpy(), ref_fm) and np.array_equal(full[3].numpy(), ref_sm)
        open(result_file, "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_partition_and_reduce(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    result = os.path.join(str(tmp_path), "result.txt")
    mp.spawn(_worker, args=(2, port, str(tmp_path), result), nprocs=2, join=True)
    assert open(result).read() == "ok"
