"""CPU tier for the N > 1 path: two gloo ranks render interleaved row shards of a frame and gather
them to rank 0; the result must equal the single-process frame bit for bit.  The render callable is
the oracle here (tests may use it); on the GPU the same shard/gather code runs over rt_render_rows
and RCCL (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, frames, out_path):
    sys.path.insert(0, ROOT)
    import importlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dpar = importlib.import_module("ray-and-pathtracer_amd.distributed")
    scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
    from oracle import oracle_api as oa
    s = oa.OracleScene()
    scenes.mixed_small(s)
    s.set_raytracer(False)
    r = oa.OracleRenderer(s, w, h)
    first, stride, count = dpar.shard_rows(h, rank, world)
    for k in range(count):  # rows first + k*stride, each as its own band of the per-pixel-seeded loop
        y = first + k * stride
        r.render(0, frames, y0=y, y1=y + 1)
    acc = torch.from_numpy(r.accumulator())
    dpar.gather_rows(acc, rank, world, 0)
    if rank == 0:
        np.save(out_path, acc.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,h", [(2, 20), (2, 21), (3, 20)])
def test_row_shard_gather_equals_single_process(world, h, tmp_path, scenes, oracle_api):
    w, frames = 24, 2
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, frames, out), nprocs=world, join=True)
    got = np.load(out)
    s = oracle_api.OracleScene()
    scenes.mixed_small(s)
    s.set_raytracer(False)
    r = oracle_api.OracleRenderer(s, w, h)
    r.render(0, frames)
    assert np.array_equal(got.view(np.uint32), r.accumulator().view(np.uint32))


def _qlearn_worker(rank, world, port, w, h, batches, frames, out_path):
    sys.path.insert(0, ROOT)
    import importlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dpar = importlib.import_module("ray-and-pathtracer_amd.distributed")
    scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
    from oracle import oracle_api as oa
    s = oa.OracleScene()
    scenes.mixed_small(s)
    s.set_raytracer(False)
    r = oa.OracleRenderer(s, w, h)
    r.qlearn_enable(6, (-4, -1, -4), (4, 5, 6), 0.3, 0.2, 1.0, 1)
    first, stride, count = dpar.shard_rows(h, rank, world)
    for b in range(batches):
        for k in range(count):
            y = first + k * stride
            r.render(b * frames, frames, y0=y, y1=y + 1)
        # the rule of DESIGN.md finding 49: the table is read-only inside a batch; the ranks' integer reward sums are added (any
        # order: integers), every rank applies the total
        sums, cnts, _ = r.qlearn_state()
        ts, tc = torch.from_numpy(sums), torch.from_numpy(cnts.astype(np.int64))
        dist.all_reduce(ts), dist.all_reduce(tc)
        r.qlearn_set_sums(ts.numpy(), tc.numpy().astype(np.uint32))
        r.qlearn_apply()
    acc = torch.from_numpy(r.accumulator())
    dpar.gather_rows(acc, rank, world, 0)
    tab = torch.from_numpy(r.qlearn_state()[2].copy())
    tabs = [torch.empty_like(tab) for _ in range(world)] if rank == 0 else None
    dist.gather(tab, tabs, dst=0)
    if rank == 0:
        for t in tabs:
            assert torch.equal(t.view(torch.int32), tab.view(torch.int32))  # every rank learned the same table
        np.save(out_path, acc.numpy())
        np.save(out_path + ".tab.npy", tab.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,h", [(2, 14), (3, 13)])
def test_qlearning_is_independent_of_the_shard(world, h, tmp_path, scenes, oracle_api):
    """The Q-learning sampler under a row shard (the oracle's statement of it; the product follows the same rule in bench.py
    --qlearn and rapt::Renderer::Tick): every rank renders its rows from the same table, the integer reward sums are all-reduced,
    every rank applies the total.  Frames and table must equal the single-process run bit for bit, batch after batch."""
    w, batches, frames = 20, 3, 2
    out = str(tmp_path / "acc.npy")
    mp.spawn(_qlearn_worker, args=(world, _free_port(), w, h, batches, frames, out), nprocs=world, join=True)
    got, got_tab = np.load(out), np.load(out + ".tab.npy")
    s = oracle_api.OracleScene()
    scenes.mixed_small(s)
    s.set_raytracer(False)
    r = oracle_api.OracleRenderer(s, w, h)
    r.qlearn_enable(6, (-4, -1, -4), (4, 5, 6), 0.3, 0.2, 1.0, 1)
    for b in range(batches):
        r.render(b * frames, frames)
        r.qlearn_apply()
    assert np.array_equal(got_tab.view(np.uint32), r.qlearn_state()[2].view(np.uint32))
    assert np.array_equal(got.view(np.uint32), r.accumulator().view(np.uint32))


class _RowPainter:
    """Stands where host_api.HostRenderer stands in distributed.render_step (the code bench.py runs per step): checks the
    row arguments exactly as rt_render_rows does (csrc/rt_api.hip: row_first >= 0, row_stride >= 1, row_count >= 1,
    last row inside the frame) and paints the rows it is asked for -- y = row_first + k * row_stride, k < row_count, the
    C ABI's rule (include/rt_amd.h) -- with a value that names the row, the frame range and how often it was painted."""

    def __init__(self, acc):
        self.acc = acc
        self.synced = 0

    def render_rows(self, mode, frame0, nframes, row_first, row_stride, row_count):
        h, w = self.acc.shape[0], self.acc.shape[1]
        assert not (row_first < 0 or row_stride < 1 or row_count < 1 or row_first + (row_count - 1) * row_stride >= h)
        for k in range(row_count):
            y = row_first + k * row_stride
            self.acc[y, :, 0] += torch.arange(w, dtype=torch.float32) + 1000.0 * y
            self.acc[y, :, 1] += float(nframes)
            self.acc[y, :, 2] += 1.0  # painted once
            self.acc[y, :, 3] = float(frame0)

    def synchronize(self):
        self.synced += 1


def _product_worker(rank, world, port, w, h, out_path):
    sys.path.insert(0, ROOT)
    import importlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dpar = importlib.import_module("ray-and-pathtracer_amd.distributed")
    acc = torch.zeros((h, w, 4), dtype=torch.float32)
    shard = dpar.RowShard(h, w, rank, world, acc.device)
    painter = _RowPainter(acc)
    for _ in range(2):  # two steps through the same preallocated buffers, like bench.py's timed loop
        acc.zero_()
        dpar.render_step(painter, acc, 1, 7, 5, shard)
    assert painter.synced == 2
    if rank == 0:
        np.save(out_path, acc.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,h", [(2, 20), (2, 21), (3, 20), (3, 24), (4, 9)])
def test_product_shard_bookkeeping(world, h, tmp_path):
    """bench.py's step (distributed.render_step: RowShard.rows() -> render_rows -> gather into preallocated buffers) with
    a renderer that enforces rt_render_rows' argument rules: after the gather rank 0 holds every row exactly once, the
    right row at the right place, for heights that do and do not divide by the number of ranks."""
    w = 13
    out = str(tmp_path / "acc.npy")
    mp.spawn(_product_worker, args=(world, _free_port(), w, h, out), nprocs=world, join=True)
    got = np.load(out)
    want = np.zeros((h, w, 4), np.float32)
    for y in range(h):
        want[y, :, 0] = np.arange(w, dtype=np.float32) + 1000.0 * y
        want[y, :, 1], want[y, :, 2], want[y, :, 3] = 5.0, 1.0, 7.0
    assert np.array_equal(got, want)


def test_shard_rows_partition():
    dpar = __import__("importlib").import_module("ray-and-pathtracer_amd.distributed")
    for h in (1, 7, 8, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            rows = []
            for r in range(world):
                f, st, c = dpar.shard_rows(h, r, world)
                rows += [f + k * st for k in range(c)]
            assert sorted(rows) == list(range(h))


def _harden_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import importlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dpar = importlib.import_module("ray-and-pathtracer_amd.distributed")
    dpar.init_group("gloo", rank, world, timeout_s=60)
    # every rank learns every rank's device, in rank order
    ids = dpar.exchange_device_ids("0000:%02x:00.0" % (0xc1 + rank // 2), world)  # ranks 0,1 share a GPU, rank 2 has its own
    assert ids == ["0000:%02x:00.0" % (0xc1 + k // 2) for k in range(world)]
    assert dpar.check_rank_devices(ids, "gloo") is False  # the rehearsal mode reports sharing ...
    try:
        dpar.check_rank_devices(ids, "nccl")  # ... an RCCL run ends on it
        raised = False
    except RuntimeError as e:
        raised = "share a GPU" in str(e)
    assert raised
    assert dpar.check_rank_devices(["a", "b", "c"][:world], "nccl") is True
    # the sampler's exchange: integer sums added in place, the same totals on every rank, counts as two's-complement int32
    sums = torch.arange(8, dtype=torch.int64) * (rank + 1) + (1 << 40)
    cnts = torch.full((8,), 0x7FFFFFF0 if rank == 0 else 0x20, dtype=torch.int64).to(torch.int32)
    dpar.all_reduce_reward_sums(sums, cnts)
    want = sum(torch.arange(8, dtype=torch.int64) * (k + 1) + (1 << 40) for k in range(world))
    assert torch.equal(sums, want)
    assert int(cnts[0].item()) & 0xFFFFFFFF == (0x7FFFFFF0 + 0x20 * (world - 1)) & 0xFFFFFFFF  # uint32 addition, bit for bit
    if rank == 0:
        open(out_path, "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_n_rank_launch_hardening(tmp_path):
    """VERDICT r5 item 5, on gloo: the device-identity exchange (one all-gather of a fixed-size byte tensor), the refusal of two ranks
    on one GPU under "nccl", the in-place all-reduce of the Q-learning reward sums."""
    out = str(tmp_path / "ok")
    mp.spawn(_harden_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    assert open(out).read() == "ok"


def _late_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import importlib, time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dpar = importlib.import_module("ray-and-pathtracer_amd.distributed")
    dpar.init_group("gloo", rank, world, timeout_s=3)
    if rank == 1:
        time.sleep(8)  # never joins the collective in time
        os._exit(0)
    t0 = time.time()
    try:
        dist.all_reduce(torch.ones(1))
        open(out_path, "w").write("completed")
    except Exception as e:  # the bounded collective raises: bench.py exits non-zero on it
        open(out_path, "w").write("raised after %.0f s" % (time.time() - t0))
    os._exit(0)


def test_a_collective_that_never_completes_ends_the_run(tmp_path):
    """A rank that does not arrive must end the run, not hang it: every collective of the process group carries the timeout of
    distributed.init_group (here 3 s on gloo; under RCCL the watchdog aborts the process)."""
    out = str(tmp_path / "res")
    mp.spawn(_late_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read().startswith("raised")


def test_nccl_needs_a_gpu_per_rank(monkeypatch):
    """`--gpus N` over RCCL on a node with fewer GPUs: refused before anything is initialised."""
    dpar = __import__("importlib").import_module("ray-and-pathtracer_amd.distributed")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit, match="need 4 GPUs"):
        dpar.init_group("nccl", 0, 4, 0)
