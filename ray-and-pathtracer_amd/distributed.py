"""Multi-GPU plumbing: one process per GPU, the frame sharded by interleaved rows, one gather of
the accumulator rows to rank 0 per reported frame (torch.distributed; backend "nccl" is RCCL over
xGMI on ROCm, "gloo" in the CPU tests).

The path shards naturally: pixels are independent (one RNG stream per pixel and frame) and the scene
is read-only, so every rank holds a full copy of the scene and renders rows rank, rank + world, ...
(rt_render_rows); interleaving balances sky rows against geometry rows.  The only exchange step is
the final framebuffer gather: W*H*16 B * (world-1)/world into rank 0, e.g. 29 MB at 1080p / 8 ranks.
"""
import torch
import torch.distributed as dist


def shard_rows(height, rank, world):
    """(row_first, row_stride, row_count) of rank's interleaved shard."""
    count = len(range(rank, height, world))
    return rank, world, count


def gather_rows(acc, rank, world, dst=0):
    """acc: [H, W, 4] float32 tensor (any device) whose rows rank::world were rendered locally.
    After the call rank dst holds every row.  Row counts may differ by one between ranks; shards are
    padded to the largest so a plain gather suffices."""
    if world == 1:
        return acc
    height = acc.shape[0]
    per = (height + world - 1) // world
    local = acc[rank::world]
    send = torch.zeros((per,) + tuple(acc.shape[1:]), dtype=acc.dtype, device=acc.device)
    send[: local.shape[0]] = local
    if rank == dst:
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, parts, dst=dst)
        for r in range(world):
            n = len(range(r, height, world))
            if r != dst:
                acc[r::world] = parts[r][:n]
    else:
        dist.gather(send, None, dst=dst)
    return acc
