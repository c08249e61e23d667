"""Image-tile data parallelism over the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

The reference splits *samples* over at most 4 devices and sums four moment planes through pinned host memory
(/root/reference/src/luminary/device/device_result_interface.c:107-299). Here the frame is split into tiles: every pixel has exactly
one owner, ranks never communicate while rendering, and the frame is assembled with ONE reduce to the display rank. Because every
sample is a pure function of (pixel, sample id), any partition reproduces the single-GPU sums bit for bit.
"""
import numpy as np


def tile_lattice_step(world):
    """The step k of the deal (== lumc_tile_lattice_step): tile (x, y) belongs to rank (x + k*y) % world; k maximises the shortest distance
    between two tiles of one rank among the steps coprime to `world` (ties: the smallest k), so a rank's tiles form the most isotropic lattice
    (8 ranks: k = 3) and the tiles a row has beyond a multiple of `world` go to every rank in turn (shares differ by at most tiles_y % world tiles)."""
    import math
    if world < 2:
        return 0
    best_k, best = 1, -1
    r = np.arange(-world, world + 1)
    a, b = np.meshgrid(r, r, indexing="xy")
    d2 = a * a + b * b
    for k in range(1, world):
        if math.gcd(k, world) != 1:
            continue
        ok = ((a + k * b) % world == 0) & (d2 > 0)
        shortest = int(d2[ok].min())
        if shortest > best:
            best, best_k = shortest, k
    return best_k


def tile_owner(tile_x, tile_y, tiles_x, world):
    """Rank of tile (tile_x, tile_y) (arrays or ints) == lumc_tile_owner. LUM_TILE_DEAL=rowmajor: round 1-4's t % world (for A/B only)."""
    import os
    if world < 2:
        return np.zeros_like(np.asarray(tile_x) + np.asarray(tile_y))
    if os.environ.get("LUM_TILE_DEAL") == "rowmajor":
        return (np.asarray(tile_y, dtype=np.int64) * tiles_x + tile_x) % world
    return (np.asarray(tile_x, dtype=np.int64) + tile_lattice_step(world) * np.asarray(tile_y, dtype=np.int64)) % world


def tile_pixels(width, height, rank, world, tile=32):
    """Pixel indices (x + y*width) owned by `rank`: its tiles (tile_owner) in row-major tile order, rows within a tile."""
    tx, ty = (width + tile - 1) // tile, (height + tile - 1) // tile
    ids = np.arange(tx * ty)
    mine = ids[tile_owner(ids % tx, ids // tx, tx, world) == rank]
    ys, xs = np.meshgrid(np.arange(tile), np.arange(tile), indexing="ij")
    px = (mine % tx)[:, None, None] * tile + xs[None]
    py = (mine // tx)[:, None, None] * tile + ys[None]
    ok = (px < width) & (py < height)
    return (px + py * width)[ok].astype(np.uint32)


def tile_share_counts(width, height, world, tile=32):
    """Pixels per rank under the deal (what every rank needs to size the gather's buffers), without building the lists."""
    tx, ty = (width + tile - 1) // tile, (height + tile - 1) // tile
    i, j = np.meshgrid(np.arange(tx), np.arange(ty), indexing="xy")
    w = np.minimum((i + 1) * tile, width) - i * tile
    h = np.minimum((j + 1) * tile, height) - j * tile
    owner = tile_owner(i, j, tx, world)
    return [int((w * h)[owner == r].sum()) for r in range(world)]


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


def gather_frame(first_moment, second_moment, width, height, rank, world, dist=None, dst=0, tile=32):
    """The same frame by a GATHER of the ranks' own pixels (what lumc_frame_gather does behind the C ABI with one ncclGather): every rank packs its planar
    accumulators into a zero-padded [4, M] buffer (M = the largest share of the tile deal, a function of (width, height, world) every rank computes),
    `dst` receives the `world` buffers and scatters each through that rank's pixel list, which it derives from the deal itself. 16 bytes per OWNED pixel
    travel instead of 16 bytes per frame pixel from every rank. Returns the [4, W*H] frame on `dst`, None elsewhere."""
    import torch
    if dist is None and world != 1:
        raise ValueError("gather_frame: %d ranks need a process group (dist=None assembles a single rank's frame only)" % world)
    shares = [tile_pixels(width, height, r, world, tile) for r in range(world)] if rank == dst else None
    if shares is not None:
        counts = [int(s.size) for s in shares]
    else:  # the share sizes follow from the deal alone
        counts = tile_share_counts(width, height, world, tile)
    stride = (max(counts) + 3) & ~3
    p = counts[rank]
    assert first_moment.numel() == 3 * p and second_moment.numel() == p, "this rank's accumulators are not its share of the deal"
    send = torch.zeros(4, stride, dtype=torch.float32, device=first_moment.device)
    send[0:3, :p] = first_moment.view(3, p)
    send[3, :p] = second_moment
    if dist is None:
        received = [send]
    else:
        received = [torch.zeros_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, received, dst=dst)
        if rank != dst:
            return None
    full = torch.zeros(4, width * height, dtype=torch.float32, device=first_moment.device)
    for r in range(world):
        idx = torch.from_numpy(np.ascontiguousarray(shares[r]).astype(np.int64)).to(full.device)
        full.index_copy_(1, idx, received[r][:, :counts[r]])
    return full


def block_mask(width, height, rank, world, tile=32):
    """Adaptive-sampling blocks (4x4 pixels, row-major over ceil(w/4) x ceil(h/4)) owned by `rank` under the same tile deal as
    tile_pixels: a block belongs to the rank of the tile it lies in (tiles are whole blocks: tile % 4 == 0)."""
    assert tile % 4 == 0
    bx, by = (width + 3) // 4, (height + 3) // 4
    tx = (width + tile - 1) // tile
    ys, xs = np.meshgrid(np.arange(by), np.arange(bx), indexing="ij")
    return (tile_owner((xs * 4) // tile, (ys * 4) // tile, tx, world) == rank).astype(np.uint8).ravel()


def adaptive_render(core, executions, dist=None, device=None):
    """Runs `executions` executions of adaptive rendering on a partitioned context (Core.adaptive_set_partition). When a stage is due,
    the ranks' block variances are summed with ONE all-reduce (every block has one owner, so the sum is a gather; 4 bytes per block)
    and every rank builds the same rates from the complete array. `dist` None = a single rank (the exchange is the identity)."""
    import torch
    remaining = executions
    while remaining > 0:
        before = sum(core.adaptive_info()["executions"])
        core.adaptive_render(remaining)
        info = core.adaptive_info()
        remaining -= sum(info["executions"]) - before
        if info["build_pending"]:
            var = torch.from_numpy(core.adaptive_variance())
            if dist is not None:
                if device is not None:
                    var = var.to(device)
                dist.all_reduce(var, op=dist.ReduceOp.SUM)
                var = var.cpu()
            core.adaptive_build_from(var.numpy())
        elif remaining > 0 and sum(info["executions"]) == before:
            raise RuntimeError("adaptive_render made no progress")
