"""Multi-GPU plumbing: one process per GPU, the frame sharded by interleaved rows, one gather of
the accumulator rows to rank 0 per reported frame (torch.distributed; backend "nccl" is RCCL over
xGMI on ROCm, "gloo" in the CPU tests).

The path shards naturally: pixels are independent (one RNG stream per pixel and frame) and the scene
is read-only, so every rank holds a full copy of the scene and renders rows rank, rank + world, ...
(rt_render_rows); interleaving balances sky rows against geometry rows.  The only exchange step is
the final framebuffer gather: W*H*16 B * (world-1)/world into rank 0, e.g. 29 MB at 1080p / 8 ranks.
xGMI is point to point, so a gather to one rank uses all of that rank's links at once; no ring.
"""
import datetime
import os
import time

import torch
import torch.distributed as dist

COLLECTIVE_TIMEOUT_S = 300  # every collective of a bench / Tick run ends within this, or the run ends with a non-zero exit code


def init_group(backend, rank, world, device_index=None, timeout_s=COLLECTIVE_TIMEOUT_S):
    """The process group of an N-rank run, made so that a rank that never arrives, or a collective that never completes, ENDS the
    run (non-zero exit) instead of hanging it: a timeout on the rendezvous and on every collective (gloo raises; for "nccl" =
    RCCL the watchdog aborts the process: TORCH_NCCL_ASYNC_ERROR_HANDLING=1), and -- "nccl" only -- one visible GPU per rank
    checked BEFORE anything is initialised.  No re-exec, no in-process restart."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
    timeout = datetime.timedelta(seconds=timeout_s)
    if backend == "nccl":
        have = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
        if have < world:
            raise SystemExit("%d ranks over RCCL need %d GPUs on this node, %d visible (RAPT_DIST_BACKEND=gloo rehearses on fewer: the ranks then share devices)" % (world, world, have))
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, timeout=timeout, device_id=torch.device("cuda", device_index))
    else:
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)


def exchange_device_ids(my_id, world, device="cpu"):
    """Every rank's device identity (a PCI address, host_api.device_pci_bus_id; any short string), in rank order, on every rank:
    one all-gather of a fixed-size byte tensor."""
    raw = my_id.encode()[:63]
    mine = torch.zeros(64, dtype=torch.uint8)
    mine[: len(raw)] = torch.tensor(list(raw), dtype=torch.uint8)
    mine = mine.to(device)
    if world == 1:
        return [my_id]
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return [bytes(p.cpu().tolist()).split(b"\0")[0].decode() for p in parts]


def check_rank_devices(ids, backend):
    """One GPU per rank is what an RCCL run means: two ranks on one device would time one GPU's work as two GPUs'.  Raises under
    "nccl"; under "gloo" (the rehearsal mode: ranks share devices on purpose) sharing is reported by the caller, not refused."""
    dup = sorted({i for i in ids if ids.count(i) > 1})
    if dup and backend == "nccl":
        raise RuntimeError("ranks share a GPU: %s (rank -> device: %s)" % (", ".join(dup), ids))
    return not dup


def all_reduce_reward_sums(sums, counts):
    """The one exchange step the Q-learning sampler adds: the ranks' pending reward sums (int64) and counts (uint32 held as int32:
    two's-complement addition is the same bits), added IN PLACE on whatever device they live on -- integers, so the order of the
    reduction does not matter and every rank ends with the same totals."""
    dist.all_reduce(sums)
    dist.all_reduce(counts)


def shard_rows(height, rank, world):
    """(row_first, row_stride, row_count) of rank's interleaved shard: the arguments of rt_render_rows."""
    count = len(range(rank, height, world))
    return rank, world, count


class RowShard:
    """The row shard of one rank and the buffers its gather needs, allocated ONCE (nothing is allocated inside a
    timed step): 'send' holds the rank's rows densely, 'staging' on the destination rank receives every rank's
    'send' ([world, rows per rank, W, 4]); the rows then go to their places in the frame with one strided copy."""

    def __init__(self, height, width, rank, world, device, dst=0, dtype=torch.float32):
        self.height, self.width, self.rank, self.world, self.dst = height, width, rank, world, dst
        self.first, self.stride, self.count = shard_rows(height, rank, world)
        self.per = (height + world - 1) // world  # rows of the largest shard: shards are padded to it, a plain gather suffices
        self.send = self.staging = self.parts = None
        if world > 1:
            self.send = torch.zeros((self.per, width, 4), dtype=dtype, device=device)
            if rank == dst:
                self.staging = torch.zeros((world, self.per, width, 4), dtype=dtype, device=device)
                self.parts = list(self.staging.unbind(0))  # contiguous views: the gather writes straight into the staging tensor

    def rows(self):
        """(row_first, row_stride, row_count) for rt_render_rows"""
        return self.first, self.stride, self.count

    def gather(self, acc):
        """acc: [H, W, 4] tensor whose rows rank::world were rendered locally.  After the call rank dst holds every row."""
        if self.world == 1:
            return acc
        self.send[: self.count].copy_(acc[self.rank :: self.world])
        if self.rank == self.dst:
            dist.gather(self.send, self.parts, dst=self.dst)
            if self.height % self.world == 0:
                # row y = k * world + r lives at staging[r, k]: one strided copy for the whole frame (the destination's own
                # rows come back unchanged from its own 'send')
                acc.view(self.per, self.world, self.width, 4).copy_(self.staging.permute(1, 0, 2, 3))
            else:
                for r in range(self.world):
                    if r != self.dst:
                        n = len(range(r, self.height, self.world))
                        acc[r :: self.world] = self.staging[r, :n]
        else:
            dist.gather(self.send, None, dst=self.dst)
        return acc


def gather_rows(acc, rank, world, dst=0):
    """One-off form of RowShard.gather (allocates its buffers; tests and small tools)."""
    if world == 1:
        return acc
    return RowShard(acc.shape[0], acc.shape[1], rank, world, acc.device, dst, acc.dtype).gather(acc)


def render_step(renderer, acc, mode, frame0, frames, shard, host_staging=None, timing=None):
    """One bench / Tick step of a rank: render the shard's rows into acc (the tensor bound as the renderer's accumulator),
    then gather them to the destination rank.  'renderer' is host_api.HostRenderer (render_rows -> rt_render_rows,
    synchronize -> rt_synchronize) or anything with those two methods.  host_staging: a pinned-size CPU tensor for the
    gloo rehearsal mode (ranks share a GPU, the gather goes through host memory); its RowShard is host_staging[1].
    timing: a dict that accumulates 'render_s' (until this rank's rows are complete) and 'gather_s' (from there until the
    exchange has completed on this rank: the wait for the slowest rank, the collective, and on the destination the strided
    copy of the staging tensor into the frame) -- with several ranks the split is what tells a slow share from a slow
    exchange.  Every step runs the same host sequence whether it is timed or not (the device is synchronised after the
    gather in warm-up steps too)."""
    t0 = time.perf_counter()
    first, stride, count = shard.rows()
    if count > 0:
        renderer.render_rows(mode, frame0, frames, first, stride, count)
    renderer.synchronize()
    t1 = time.perf_counter()
    if host_staging is None:
        shard.gather(acc)
        if shard.world > 1 and acc.is_cuda:
            torch.cuda.synchronize(acc.device)  # the collective is asynchronous on the device: its end is the step's end
    else:
        host, host_shard = host_staging
        host.copy_(acc)
        host_shard.gather(host)
        if shard.rank == shard.dst:
            acc.copy_(host)
    if timing is not None:
        timing["render_s"] = timing.get("render_s", 0.0) + (t1 - t0)
        timing["gather_s"] = timing.get("gather_s", 0.0) + (time.perf_counter() - t1)
    return acc
