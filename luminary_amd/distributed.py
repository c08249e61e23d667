"""Image-tile data parallelism over the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

The reference splits *samples* over at most 4 devices and sums four moment planes through pinned host memory
(/root/reference/src/luminary/device/device_result_interface.c:107-299). Here the frame is split into tiles: every pixel has exactly
one owner, ranks never communicate while rendering, and the frame is assembled with ONE reduce to the display rank. Because every
sample is a pure function of (pixel, sample id), any partition reproduces the single-GPU sums bit for bit.
"""
import numpy as np


def tile_pixels(width, height, rank, world, tile=32):
    """Pixel indices (x + y*width) owned by `rank`: tile t of the row-major tile grid belongs to rank t % world."""
    tx, ty = (width + tile - 1) // tile, (height + tile - 1) // tile
    ids = np.arange(tx * ty)
    mine = ids[ids % world == rank]
    ys, xs = np.meshgrid(np.arange(tile), np.arange(tile), indexing="ij")
    px = (mine % tx)[:, None, None] * tile + xs[None]
    py = (mine // tx)[:, None, None] * tile + ys[None]
    ok = (px < width) & (py < height)
    return (px + py * width)[ok].astype(np.uint32)


def assemble_frame(first_moment, second_moment, pixels, num_frame_pixels, dist=None, dst=0):
    """Scatters this rank's planar accumulators ([3*P] and [P] torch tensors) into a zero [4, W*H] frame and reduces it to `dst`.
    Every pixel is written by exactly one rank, so the SUM is a gather; one collective of 16 bytes per pixel per output."""
    import torch
    full = torch.zeros(4, num_frame_pixels, dtype=torch.float32, device=first_moment.device)
    idx = torch.from_numpy(np.ascontiguousarray(pixels).astype(np.int64)).to(first_moment.device)
    p = idx.numel()
    full[0:3].index_copy_(1, idx, first_moment.view(3, p))
    full[3].index_copy_(0, idx, second_moment)
    if dist is not None:
        dist.reduce(full, dst=dst, op=dist.ReduceOp.SUM)
    return full
