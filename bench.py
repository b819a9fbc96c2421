#!/usr/bin/env python3
"""bench.py -- Mrays/s of the trace loop on MI355X for the BASELINE.json workload.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One *step* = one complete render of the workload frame: 1920x1080 pixels x 64 frames (spp) of the
path integrator over BASELINE config 3 (unity.tri as two BLASes, 8 instances through the TLAS =
100,672 instanced triangles, glass + metal, 4 spheres, floor plane, 2 area lights), scene and
accumulator resident in HBM.  With N ranks the frame's rows are interleaved over the ranks and the
accumulator rows are gathered to rank 0 inside the step (strong scaling: total work fixed).

Rank 0 prints ONE JSON line.  value = W*H*spp / seconds / 1e6, the reference's own definition of
"Mrays/s" (renderer.cpp:300-304: primary pixel samples per second); all traced rays per second are
reported beside it.  roofline: dominant kernel = k_extend_s (Scene::FindNearest), see roofline_block():
what binds it on this workload (the vector-memory path: texture addressers + L1; the scene lives in
L2), with the SURVEY.md 8(d) algorithmic-bytes rate (work counters from an untimed counting pass of
the identical, deterministic workload / device time measured with HIP events on the kernel's stream
during the timed steps), the counter-measured HBM rate and the VALU figure beside it -- and, measured
LIVE by this run outside the timed region (out_of_cache_leg), the same kernel on a scene the caches
cannot hold (8.4 M-triangle terrain, built on the device): there SURVEY 8(d)'s bytes over the HBM
peak is a roofline fraction (<= 1).  tick_ms / share_ms: Renderer::Tick latency (Whitted, path) and
the eight 1/8 row shares of the headline step, live as well.  cpu_baseline: the oracle (oracle/,
the CPU restatement; kind "port") on the host cores this job may use and on one thread, on bounded
samples; parity_check: the oracle's accumulator of those frames against the device's, the whole
frame (rc != 0 above 1e-4); the out-of-cache scene's crop of hits against the oracle's (rc != 0 when
a bit differs).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def pkg(name):
    return importlib.import_module("ray-and-pathtracer_amd." + name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="config3", choices=["config2", "config3", "config4", "config5"])
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-count", action="store_true", help="profiling runs: skip the untimed counting pass (ray counts and algorithmic bytes are then 0)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--qlearn", type=int, default=0, metavar="FRAMES", help="Q-learning guided sampling (rt_qlearn_*, Dahm & Keller 2017; the reference has no code for it: "
                    "PARITY UNPINNED): the step renders its spp in batches of FRAMES frames and folds the rewards into the table between them "
                    "(with N ranks the integer reward sums are all-reduced first, so every rank learns the same table)")
    ap.add_argument("--qlearn-mask", type=int, default=3, help="rt_qlearn_params::learn_mask: 3 = every fourth sample pays rewards (all samples pick guided), 0 = all")
    ap.add_argument("--emulate-world", type=int, default=0, help="profiling on ONE GPU: render only the rows rank 0 of an N-rank run renders "
                    "(no process group, no gather): the counters of that share are what rank 0 of the N-GPU run is priced with")
    ap.add_argument("--emulate-rank", type=int, default=0, help="with --emulate-world N: the rows of rank R instead of rank 0's (the N-GPU step is as "
                    "long as its SLOWEST share: profiles/r04_shares_all_ranks.txt)")
    ap.add_argument("--no-legs", action="store_true", help="skip the live legs outside the timed region (out-of-cache terrain, Tick latency, 1/8 shares); --no-cpu-baseline skips them too")
    ap.add_argument("--force-legs", action="store_true", help="tests: run the live legs whatever the workload's size")
    ap.add_argument("--ooc-n", type=int, default=2048, help="out-of-cache leg: the terrain is 2 n^2 triangles (2048: 8.4 M, pairs + primitive records 0.94 GB)")
    ap.add_argument("--ooc-spp", type=int, default=16)
    ap.add_argument("--ooc-steps", type=int, default=3)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: start the N ranks as a CHILD process (torch.distributed.run, one rank per GPU) before
        # torch is imported or HIP touched in this one, relay rank 0's JSON line (the ranks inherit stdout) and the exit code
        # (the launcher picks the rendezvous port itself -- endpoint port 0 -- : no bind-then-close race with other jobs on the box)
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--rdzv-backend=c10d",
               "--rdzv-endpoint=127.0.0.1:0", "--local-addr", "127.0.0.1", os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; this benchmark has no CPU path")
    # RAPT_DIST_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: ranks share
    # devices (local_rank % device_count) and the gather goes through host memory.  The driver's
    # multi-GPU runs use the default: one GPU per rank, "nccl" (= RCCL over xGMI).
    backend = os.environ.get("RAPT_DIST_BACKEND", "nccl")
    ha, scenes, dpar = pkg("host_api"), pkg("scenes"), pkg("distributed")
    if world > 1 and backend == "nccl" and torch.cuda.device_count() < world:  # before anything is initialised: a rank per GPU or no run
        raise SystemExit("bench.py: --gpus %d over RCCL needs %d GPUs, %d visible" % (world, world, torch.cuda.device_count()))
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    if world > 1:
        dpar.init_group(backend, rank, world, device_index)  # timeouts on the rendezvous and on every collective: a hang becomes a non-zero exit
    # one rank (re)builds stale libraries; the others wait, they would race in the same directory
    if rank == 0:
        ha.build()
    if world > 1:
        dist.barrier()
    # which GPU every rank really sits on (PCI address through the C ABI), on every rank; two ranks on one GPU end an RCCL run here
    red_dev0 = "cuda" if backend == "nccl" else "cpu"
    ranks_devices = dpar.exchange_device_ids(ha.device_pci_bus_id(device_index), world, red_dev0)
    try:
        devices_distinct = dpar.check_rank_devices(ranks_devices, backend if world > 1 else "single")
    except RuntimeError as e:
        raise SystemExit("bench.py: %s" % e)

    # ---- workload ----
    probe = ha.HostScene()
    cfg = scenes.REGISTRY[args.workload](probe)
    probe.close()
    W = args.width or cfg["width"]
    H = args.height or cfg["height"]
    spp = args.spp or cfg["frames"]
    r = ha.HostRenderer(W, H, device_index)
    scenes.REGISTRY[args.workload](r.scene)
    r.commit()
    if "camera" in cfg:
        c = cfg["camera"]
        r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
    mode = ha.RT_MODE_PATH
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    r.bind_accumulator(acc.data_ptr())
    red_dev_early = "cuda" if backend == "nccl" else "cpu"
    # the shard of this rank and every buffer its gather needs: allocated once, outside the timed region
    shard = dpar.RowShard(H, W, rank, world, acc.device)
    if args.emulate_world > 1:
        if world != 1:
            raise SystemExit("bench.py: --emulate-world runs on one rank")
        shard = dpar.RowShard(H, W, 0, 1, acc.device)
        if not 0 <= args.emulate_rank < args.emulate_world:
            raise SystemExit("bench.py: --emulate-rank %d of %d" % (args.emulate_rank, args.emulate_world))
        shard.first, shard.stride, shard.count = dpar.shard_rows(H, args.emulate_rank, args.emulate_world)
    host_staging = None
    if world > 1 and backend != "nccl":
        host_staging = (torch.zeros((H, W, 4), dtype=torch.float32), dpar.RowShard(H, W, rank, world, torch.device("cpu")))

    qbox = cfg.get("qbox", ((-12.0, -2.0, -8.0), (12.0, 10.0, 16.0)))
    timing, timed = {}, [False]  # render / gather split of the timed steps (several ranks)
    # several ranks with the sampler on: the pending reward sums live in torch tensors on the device (rt_qlearn_bind_sums) and are
    # all-reduced in place -- RCCL on device memory, no host copy in the exchange (the gloo rehearsal stages through the host)
    qsum = qcnt = None
    if args.qlearn and world > 1:
        qsum = torch.zeros(16 ** 3 * 64, dtype=torch.int64, device="cuda")
        qcnt = torch.zeros(16 ** 3 * 64, dtype=torch.int32, device="cuda")

    def step():
        acc.zero_()
        torch.cuda.synchronize()
        if not args.qlearn:
            dpar.render_step(r, acc, mode, 0, spp, shard, host_staging, timing if timed[0] else None)
            return
        # every step learns from scratch, so that the K timed steps do the same work
        r.qlearn_enable(16, qbox[0], qbox[1], 0.3, 0.2, 1.0, args.qlearn_mask)
        if qsum is not None:
            r.qlearn_bind_sums(qsum.data_ptr(), qcnt.data_ptr())
        first, stride, count = shard.rows()
        for f0 in range(0, spp, args.qlearn):
            r.render_rows(mode, f0, min(args.qlearn, spp - f0), first, stride, count)  # (returns with the batch complete and its rewards folded into the sums)
            if world > 1:  # the one exchange step this sampler adds: integer sums, so the order of the reduction does not matter
                if red_dev_early == "cuda":
                    dpar.all_reduce_reward_sums(qsum, qcnt)
                    torch.cuda.current_stream().synchronize()  # the apply below runs on the renderer's own stream
                else:
                    hs, hc = qsum.cpu(), qcnt.cpu()
                    dpar.all_reduce_reward_sums(hs, hc)
                    qsum.copy_(hs), qcnt.copy_(hc)
                    torch.cuda.synchronize()
            r.qlearn_apply()
        r.synchronize()
        if host_staging is None:
            shard.gather(acc)
        else:
            host_staging[0].copy_(acc)
            host_staging[1].gather(host_staging[0])
            if rank == 0:
                acc.copy_(host_staging[0])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        r.synchronize()

    # ---- untimed: counting pass (algorithmic work of this rank's rows), then warmup ----
    r.counters()
    if not args.no_count:
        r.set_counting(ha.RT_COUNT_EXECUTED)  # the walk the timed kernels make (same results, fewer TLAS / instance visits than the reference's)
        step()
    near, occl = r.counters_split()
    r.set_counting(False)
    for _ in range(args.warmup):
        step()

    # ---- timed region: exactly K steps ----
    r.set_profiling(True)
    r.profile()
    fence()
    timed[0] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    timed[0] = False
    prof = r.profile()
    r.set_profiling(False)

    red_dev = "cuda" if backend == "nccl" else "cpu"
    t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    cnt = torch.tensor([near["rays_nearest"], occl["rays_occluded"]], dtype=torch.float64, device=red_dev)
    split = torch.tensor([timing.get("render_s", 0.0), timing.get("gather_s", 0.0)], dtype=torch.float64, device=red_dev)
    split0 = split.clone()
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        dist.all_reduce(split, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    rays_all = float(cnt.sum().item())  # per step, all ranks
    seen = torch.ones(1, dtype=torch.int64, device=red_dev)  # the witness that the process group really had N ranks
    if world > 1:
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)

    failures = []
    if rank == 0:
        sec_per_step = dt / args.steps
        value = W * H * spp / sec_per_step / 1e6
        if args.emulate_world > 1:  # a profiling line: only rank 0's rows of the N-rank shard were rendered
            value = W * shard.count * spp / sec_per_step / 1e6
        ext = prof["extend"]
        launches_per_step = ext["launches"] / args.steps
        avg_ms = ext["ms"] / max(1, ext["launches"])  # HIP events on the kernel's own stream, inside the timed steps
        out = {
            "metric": ("Mrays/s at %d×%d×%dspp" % (W, H, spp)) + (" (rank %d's rows of a %d-rank shard only: a profiling line)" % (args.emulate_rank, args.emulate_world) if args.emulate_world > 1 else ""),
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "ranks_seen": int(seen.item()), "backend": backend if world > 1 else None,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(sec_per_step * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s, %dx%d, %d spp, path integrator" % (args.workload, cfg["name"], W, H, spp),
                       "parallelism": "row-interleaved pixel shard x%d + accumulator gather to rank 0" % world,
                       "sampler": ("Q-learning guided indirect bounce (Dahm & Keller 2017; no reference code: PARITY UNPINNED), 16^3 cells x 64 patches, "
                                   "table updated every %d frames, rewards from samples with stream state & %d == 0" % (args.qlearn, args.qlearn_mask)) if args.qlearn else "uniform hemisphere (renderer.cpp:181)",
                       "rays_definition": "value counts primary pixel samples (reference's Mrays/s, renderer.cpp:300); all_rays counts every FindNearest + IsOccluded query"},
            "all_rays_mrays_per_s": round(rays_all / sec_per_step / 1e6, 3),
            "rays_per_step": {"nearest": int(cnt[0].item()), "occluded": int(cnt[1].item())},
            # several ranks: where a step's time goes on the host's clock -- render = until the rank's own rows are complete, gather =
            # from there until the exchange has completed on the rank (it includes waiting for the slowest share); rank 0 and the
            # maximum over the ranks.  null on one rank (nothing is gathered) and under --qlearn (the step is several batches).
            "render_ms": ({"rank0": round(float(split0[0].item()) / args.steps * 1e3, 3), "max": round(float(split[0].item()) / args.steps * 1e3, 3)} if world > 1 and not args.qlearn else None),
            "gather_ms": ({"rank0": round(float(split0[1].item()) / args.steps * 1e3, 3), "max": round(float(split[1].item()) / args.steps * 1e3, 3)} if world > 1 and not args.qlearn else None),
            "roofline": roofline_block(args, ha, near, occl, avg_ms, launches_per_step, W, H, spp, args.emulate_world if args.emulate_world > 1 else world,
                                       {k: round(v["ms"] / args.steps, 3) for k, v in prof.items() if v["launches"]},
                                       r.build_info(), sec_per_step, prof["connect"]),
        }
        out["frame_checksum"] = frame_checksum(acc)
        out["ranks_devices"] = ranks_devices  # PCI address of every rank's GPU, in rank order (one all-gather; distinct under RCCL or the run has ended above)
        out["ranks_devices_distinct"] = bool(devices_distinct)
        # several ranks: the gathered frame must be the one-GPU frame bit for bit (rows are independent, sums are in frame order); the
        # committed one-GPU line of the same workload is the witness (its own frame was compared with the oracle's: parity_check there)
        try:
            ref = json.loads(open(os.path.join(ROOT, "profiles", "final_bench.json")).read().strip().splitlines()[-1])
            if ref["metric"] == out["metric"] and not args.qlearn and args.emulate_world <= 1:
                out["frame_checksum_of_committed_1gpu_line"] = ref["frame_checksum"]
                out["frame_equals_committed_1gpu_frame"] = ref["frame_checksum"] == out["frame_checksum"]
                out["committed_1gpu_line_is_of_these_kernels"] = ref["roofline"]["pmc"]["kernel_hash"] == out["roofline"]["pmc"]["kernel_hash"]
        except Exception:
            pass
        if int(seen.item()) != world:
            failures.append("the process group held %d ranks, not %d" % (int(seen.item()), world))
        if world > 1 and out.get("frame_equals_committed_1gpu_frame") is False and out.get("committed_1gpu_line_is_of_these_kernels"):  # same kernels, another frame: the shard or the gather is wrong
            failures.append("the gathered %d-rank frame is not the committed one-GPU frame (%s vs %s)" % (world, out["frame_checksum"], out["frame_checksum_of_committed_1gpu_line"]))
        # ---- live legs outside the timed region (one GPU, the default workload): what the driver cannot see otherwise ----
        plain = world == 1 and not args.qlearn and args.emulate_world <= 1 and ((args.workload == "config3" and not (args.width or args.height or args.spp)) or args.force_legs)
        terrain = None
        if plain and not args.no_legs and not args.no_cpu_baseline:  # (profiling runs pass --no-cpu-baseline: their kernel statistics are the timed steps' alone)
            t_legs = time.perf_counter()
            out["share_ms"] = share_leg(dpar, r, acc, mode, spp, H, W, sec_per_step * 1e3)
            out["tick_ms"] = tick_leg(ha, scenes, device_index)
            ooc, terrain = out_of_cache_leg(args, ha, scenes, device_index)
            out["roofline"]["hbm"]["out_of_cache_live"] = ooc
            out["gpu_leg_s"] = round(time.perf_counter() - t_legs, 2)
        if world == 1 and not args.no_cpu_baseline:
            t_cpu = time.perf_counter()
            can_check = not args.qlearn and args.emulate_world <= 1
            out["cpu_baseline"], out["parity_check"] = cpu_baseline(args, cfg, W, H, spp, (r, acc, mode, out["frame_checksum"]) if can_check else None)
            if out["parity_check"] is not None and not out["parity_check"]["ok"]:
                failures.append("the frame differs from the oracle's beyond the tolerance: %s" % json.dumps(out["parity_check"]))
            if terrain is not None:
                crop = out_of_cache_crop_check(args, scenes, terrain)
                out["roofline"]["hbm"]["out_of_cache_live"]["crop_parity"] = crop
                if not crop["bit_exact"]:
                    failures.append("out-of-cache scene: the crop's hits differ from the oracle's: %s" % json.dumps(crop))
            out["cpu_leg_s"] = round(time.perf_counter() - t_cpu, 2)
        if terrain is not None:
            terrain[0].close()
        out["timed_region_s"] = round(dt, 3)
        print(json.dumps(out), flush=True)
    r.close()
    if world > 1:  # (rank 0 leaves through the same barrier as the others before it reports a failure: nobody is left waiting for it)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and failures:
        raise SystemExit("bench.py: " + "; ".join(failures))


def frame_checksum(acc, rows=None):
    """64-bit sum of the accumulator's float bits (of the rows first::stride when given): equal frames, equal sums"""
    import torch
    a = acc if rows is None else acc[rows[0]::rows[1]]
    return "%016x" % int(torch.sum(a.contiguous().view(torch.int32).to(torch.int64)).item() & 0xFFFFFFFFFFFFFFFF)


def share_leg(dpar, r, acc, mode, spp, H, W, full_ms, world=8, steps=3):
    """The eight interleaved 1/8 row shares of the headline step, each rendered alone on this GPU (what rank k of an 8-GPU run
    renders; no process group, no gather): an N-GPU step is as long as its slowest share, so full / max is the speed-up the row
    shard can reach before the exchange -- a projection from one GPU, not a multi-GPU measurement."""
    import torch
    ms, sums = [], []
    for k in range(world):
        shard = dpar.RowShard(H, W, 0, 1, acc.device)
        shard.first, shard.stride, shard.count = dpar.shard_rows(H, k, world)
        best = None
        for it in range(steps + 1):  # the first one warms the share's state sizes up
            acc.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dpar.render_step(r, acc, mode, 0, spp, shard)
            dt = (time.perf_counter() - t0) * 1e3
            if it > 0:
                best = dt if best is None else min(best, dt)
        ms.append(round(best, 3))
        sums.append(frame_checksum(acc, (k, world)))
    return {"world": world, "per_rank": ms, "slowest": max(ms), "mean": round(sum(ms) / len(ms), 3), "steps": steps, "statistic": "best of %d" % steps,
            "full_step_ms": round(full_ms, 3), "projected_speedup": round(full_ms / max(ms), 3), "row_set_checksums": sums,
            "note": "projection from ONE GPU (each share alone on it; no exchange): not a multi-GPU measurement"}


def tick_leg(ha, scenes, device_index, ticks=20):
    """rapt::Renderer::Tick (renderer.cpp:240-305: one frame per call, pixels resolved to the host every call) at 1920x1080 on the
    headline scene, Whitted and path mode: host-clock milliseconds per Tick, the accumulator's host mirror off (the PCIe-inclusive
    figure with it on is in profiles/)."""
    out = {"scene": "config3's (pretty_tlas, 8 instances), 1920x1080", "ticks": ticks, "includes": "rt_render + rt_resolve + the 8-MB pixel download of every Tick"}
    for name, path in (("whitted", False), ("path", True)):
        r = ha.HostRenderer(1920, 1080, device_index)
        d = scenes.REGISTRY["config3"](r.scene)
        r.commit()
        c = d["camera"]
        r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
        r.scene.set_raytracer(not path)
        r.L.rth_renderer_set_download(r.h, 0)
        for _ in range(6):
            r.tick()
        t0 = time.perf_counter()
        for _ in range(ticks):
            r.tick()
        out[name] = round((time.perf_counter() - t0) / ticks * 1e3, 3)
        px = r.tick_pixels()
        out[name + "_pixels_checksum"] = "%016x" % (int(px.astype("uint64").sum()) & 0xFFFFFFFFFFFFFFFF)
        r.close()
    return out


def out_of_cache_leg(args, ha, scenes, device_index):
    """The regime north_star's "fraction of the HBM-read roofline on BVH traversal" is defined in: a scene the caches cannot hold.
    scenes.terrain_scene: 2 n^2 triangles (n = 2048: 8.4 M, pair + primitive records 0.94 GB = several L2s + Infinity Caches), ONE
    scene BVH built with rt_build_bvh_split ON THE DEVICE (binned SAH, the reference's tree bit for bit), rendered 1920x1080 x spp
    in path mode: a counting pass gives SURVEY 8(d)'s bytes for exactly these rays (64 B per inner visit, 52 per primitive test,
    48 per ray), HIP events on the kernel's stream give k_extend_s's time over the timed steps.  -> (record, (renderer, scene
    description)); the renderer stays open for the crop check of the CPU leg."""
    W, H, spp, steps = 1920, 1080, args.ooc_spp, args.ooc_steps
    t0 = time.perf_counter()
    r = ha.HostRenderer(W, H, device_index)
    r.scene.device_build(r.ctx)  # Scene::BuildBVH -> rt_build_bvh_split
    d = scenes.terrain_scene(r.scene, n=args.ooc_n)
    t_build = time.perf_counter() - t0
    r.commit()
    t_commit = time.perf_counter() - t0 - t_build
    c = d["camera"]
    r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
    info = r.scene.bvh_dump_info()
    pair_bytes, prim_bytes = info["nodes_used"] // 2 * 64, info["N"] * 64
    r.set_counting(ha.RT_COUNT_EXECUTED)
    r.counters()
    r.clear(), r.render(ha.RT_MODE_PATH, 0, spp)
    near, occl = r.counters_split()
    r.set_counting(False)
    r.clear(), r.render(ha.RT_MODE_PATH, 0, spp), r.synchronize()
    r.set_profiling(True)
    r.profile()
    t1 = time.perf_counter()
    for _ in range(steps):
        r.clear(), r.render(ha.RT_MODE_PATH, 0, spp)
    r.synchronize()
    dt = (time.perf_counter() - t1) / steps
    pr = r.profile()
    r.set_profiling(False)
    ext, con = pr["extend"], pr["connect"]
    ext_ms, con_ms = ext["ms"] / max(1, ext["launches"]), con["ms"] / max(1, con["launches"])
    b_ext, b_con = ha.algorithmic_bytes(near), ha.algorithmic_bytes(occl)
    per_launch = b_ext / max(1.0, ext["launches"] / steps)
    rec = {"scene": "terrain n=%d: %d triangles, %d BVH nodes, pair + primitive records %.3f GB; built by rt_build_bvh_split on the device (binned SAH)" % (args.ooc_n, d["triangles"], info["nodes_used"], (pair_bytes + prim_bytes) / 1e9),
           "scene_bytes": pair_bytes + prim_bytes, "frame": "%dx%d x %d spp, path integrator" % (W, H, spp), "steps": steps,
           "build_s": round(t_build, 2), "upload_s": round(t_commit, 2), "ms_per_step": round(dt * 1e3, 3),
           "kernel": "k_extend_s<false>", "kernel_ms": round(ext_ms, 4), "launches_per_step": ext["launches"] / steps,
           "algorithmic_bytes_per_launch": int(per_launch), "algorithmic_GBps": round(per_launch / (ext_ms * 1e-3) / 1e9, 1),
           "algorithmic_over_hbm_peak": round(per_launch / (ext_ms * 1e-3) / 8e12, 4), "peak_GBps": 8000.0,
           "traffic": None, "hbm_GBps": None, "frac_of_hbm_peak": None,
           "connect_kernel_ms": round(con_ms, 4), "connect_algorithmic_GBps": round(b_con / max(1.0, con["launches"] / steps) / (con_ms * 1e-3) / 1e9, 1) if con_ms > 0 else None,
           "work_per_step": {k: int(near[k]) for k in ("inner_visits", "prim_tests", "rays_nearest")},
           "kernel_ms_per_step": {k: round(v["ms"] / steps, 3) for k, v in pr.items() if v["launches"]},
           "note": "algorithmic_* = SURVEY 8(d) bytes of the rays actually traced / HIP-event kernel time: it counts every node pair and primitive a ray touches, and the top levels of the tree "
                   "(~40 % of the requests) are L2 hits even here, so it can exceed the HBM peak; frac_of_hbm_peak = traffic (HBM bytes per launch from the PMC counters of this kernel on this scene, "
                   "profiles/out_of_cache_pmc.json, quoted only while measured on these kernel sources) / the same live kernel time / 8 TB/s: the roofline fraction"}
    src_hash = kernel_hash(r.build_info().split(" | ")[0])
    try:  # the counter side (rocprofv3 --pmc of profiles/out_of_cache.py: HBM bytes, L2 hit rate, TA busy), quoted only while it is these kernels'
        pj = json.load(open(os.path.join(ROOT, "profiles", "out_of_cache_pmc.json")))
        if pj.get("kernel_hash") == src_hash:
            k = pj["kernels"]["k_extend_s<false>"]
            traffic = k["hbm_GB"] * 1e9 / k["launches"]  # per launch: the rays are the same in every run
            rec["traffic"] = int(traffic)
            rec["hbm_GBps"] = round(traffic / (ext_ms * 1e-3) / 1e9, 1)
            rec["frac_of_hbm_peak"] = round(traffic / (ext_ms * 1e-3) / 8e12, 4)
            rec["frac_of_measured_stream_peak"] = round(traffic / (ext_ms * 1e-3) / 6.29e12, 4)  # 6.29 TB/s: the streaming-read rate MI355X_MICROARCH.md measures on this part: the practical ceiling beside the data sheet's 8 TB/s (SURVEY 8d asks for both)
            rec["counters"] = {"file": "profiles/out_of_cache_pmc.json", "used": True, "profiled_launch_ms": round(k["ms_total"] / k["launches"], 4), "profiled_hbm_TBps": k["hbm_TBps"],
                               "profiled_frac_of_hbm_peak": k["frac_of_8TBps"], "l2_hit_rate": k["l2_hit_rate"], "ta_busy_avg": k["ta_busy_avg"], "ta_busy_max": k["ta_busy_max"]}
        else:
            rec["counters"] = {"file": "profiles/out_of_cache_pmc.json", "used": False, "why": "measured on other kernel sources (%s, these are %s)" % (pj.get("kernel_hash"), src_hash)}
    except Exception:
        rec["counters"] = None
    return rec, (r, d)


def out_of_cache_crop_check(args, scenes, terrain):
    """CPU leg: the oracle builds the same terrain and the same tree on the host and answers the primary rays of a 64 x 64 crop of
    the frame (below the horizon); the device's hit ids, distances, normals and occlusion flags for the same rays must equal them
    bit for bit (the witness that the out-of-cache figures were measured on the reference's traversal)."""
    import numpy as np
    from oracle import oracle_api as oa
    r, d = terrain
    W, H, c = 1920, 1080, d["camera"]
    t0 = time.perf_counter()
    o = oa.OracleScene()
    scenes.terrain_scene(o, n=args.ooc_n)
    tl, tr, bl = (np.array(c[k], np.float64) for k in ("top_left", "top_right", "bottom_left"))
    x0, y0, n = 928, 560, 64
    P = lambda x, y: tl + (x / W) * (tr - tl) + (y / H) * (bl - tl)
    orr = oa.OracleRenderer(o, n, n)
    orr.set_camera(c["cam_pos"], tuple(P(x0, y0)), tuple(P(x0 + n, y0)), tuple(P(x0, y0 + n)))
    O, D = orr.primary_rays()
    ref, got = o.find_nearest(O, D, t_min=1e-6), r.find_nearest(O, D, t_min=1e-6)
    hit = ref["obj"] != -1
    ok = (np.array_equal(got["obj"], ref["obj"]) and np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))
          and np.array_equal(got["normal"][hit].view(np.uint32), ref["normal"][hit].view(np.uint32)))
    ok = ok and np.array_equal(o.is_occluded(O, D)["occluded"], r.is_occluded(O, D))
    orr.close()
    o.close()
    return {"crop": [x0, y0, n, n], "rays": int(len(O)), "hits": int(hit.sum()), "bit_exact": bool(ok), "oracle_s": round(time.perf_counter() - t0, 1)}


def kernel_hash(build_info=""):
    """Hash of everything that decides how many instructions / bytes a launch issues: the device sources, rt_api.hip
    (round loop, launch geometry, tuning defaults), the Makefile, and the library's own account of its compile flags
    (EXTRA=...) and of the tuning the context resolved from the environment (rt_build_info / rt_tuning_info).  Counter
    files measured on anything else are ignored (profiles/valu_roofline.py stamps the same hash; .git does not travel to
    the GPU box, so a commit id cannot be used)."""
    import hashlib
    d = os.path.join(ROOT, "ray-and-pathtracer_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith(".h") or f.endswith(".hip") or f.endswith(".inc") or f == "Makefile":
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    h.update(build_info.encode())
    return h.hexdigest()[:16]


def roofline_block(args, ha, near, occl, avg_ms, launches_per_step, W, H, spp, world, kernel_ms, build_info, sec_per_step, prof_connect):
    """Dominant kernel k_extend_s (Scene::FindNearest).  What binds it ON THIS WORKLOAD, measured: neither HBM
    (the scene is a few MB and lives in L2: counter HBM traffic ~0.1 of the peak) nor MFMA (none on this path) nor the VALU
    (enabled-lane issue slots ~0.25 of the peak) but the vector-memory path of the CUs: every lane's 64-byte record is a separate
    access through the texture addresser (TA) and L1.  So:
      bound    = "vector-memory (TA/L1)"
      achieved = TA busy fraction of the launch: TA busy cycles per launch (avg over the TAs; from the committed PMC summary of THESE
                 kernel sources -- the workload is deterministic, the same accesses are made in every run) / the launch duration
                 measured LIVE with HIP events on the kernel's stream inside the timed steps
      peak     = 1.0 (a TA busy every cycle), frac = achieved / peak
      traffic  = HBM bytes per launch from the PMC counters (FETCH_SIZE + WRITE_SIZE, the guide's gfx950 units)
    'sides' puts every other candidate ceiling beside it, each recomputable from profiles/roofline_pmc.json + the live time: L1
    accesses per CU-cycle, the enabled-lane VALU fraction, the counter HBM fraction, and SURVEY 8(d)'s algorithmic-bytes demand over
    the HBM peak -- which exceeds 1 here because the bytes are served by L1/L2, i.e. it is a demand figure, not a roofline; the
    regime where it IS a roofline (a scene the caches cannot hold) is measured live by this run: hbm.out_of_cache_live.
    Counter-derived fields are null when profiles/roofline_pmc.json was measured on other kernel sources, on another workload size,
    or with another number of ranks."""
    bytes_extend = ha.algorithmic_bytes(near, executed=True)  # this rank, one step: 64 I + 52 P + 48 R + 64 T + 128 X
    reach_bytes = 48 * near["tlas_inner"]  # the builder's own reach[] records, reported apart from SURVEY 8(d)'s terms
    sec = avg_ms * 1e-3
    alg_per_launch = bytes_extend / max(1.0, launches_per_step)
    rb = {"bound": "vector-memory (TA/L1)", "kernel": "k_extend_s<false> (Scene::FindNearest)", "achieved": None, "peak": 1.0, "unit": "TA busy fraction",
          "frac": None, "traffic": None,
          "avg_launch_ms": round(avg_ms, 5), "launches_per_step": launches_per_step, "kernel_ms_per_step": kernel_ms,
          "sides": None,
          "hbm": {"peak_GBps": 8000.0,
                  "algorithmic_bytes_per_launch": int(alg_per_launch), "reach_bytes_per_launch": int(reach_bytes / max(1.0, launches_per_step)),
                  "algorithmic_GBps": round(alg_per_launch / sec / 1e9, 2) if sec > 0 else None,
                  "algorithmic_demand_over_hbm_peak": round(alg_per_launch / sec / 8e12, 4) if sec > 0 else None,
                  "note": "served by caches: SURVEY 8(d)'s algorithmic bytes (every node pair and primitive a ray touches, as if fetched from memory) exceed what HBM could deliver because this scene is served from L1/L2 -- a demand figure, not a roofline; see out_of_cache_live for the regime where it is one"},
          "algorithmic_work_per_step": {k: int(near[k]) for k in ("inner_visits", "prim_tests", "tlas_inner", "instance_visits", "rays_nearest")}}
    # the same SURVEY 8(d) figure for the any-hit kernel and for the whole step (both traversal kernels' bytes over the step time;
    # rank 0's rows when the frame is sharded)
    bytes_connect = ha.algorithmic_bytes(occl, executed=True)
    con_launches = max(1.0, prof_connect["launches"] / max(1, args.steps))
    con_sec = prof_connect["ms"] / max(1, prof_connect["launches"]) * 1e-3
    rb["hbm"]["connect_algorithmic_bytes_per_launch"] = int(bytes_connect / con_launches)
    rb["hbm"]["connect_algorithmic_GBps"] = round(bytes_connect / con_launches / con_sec / 1e9, 2) if con_sec > 0 else None
    rb["hbm"]["step_algorithmic_bytes"] = int(bytes_extend + bytes_connect)
    rb["hbm"]["step_algorithmic_GBps"] = round((bytes_extend + bytes_connect) / sec_per_step / 1e9, 2) if sec_per_step > 0 else None
    rb["hbm"]["step_algorithmic_demand_over_hbm_peak"] = round((bytes_extend + bytes_connect) / sec_per_step / 8e12, 4) if sec_per_step > 0 else None
    rb["build"] = build_info
    ppath = os.path.join(ROOT, "profiles", "roofline_pmc.json")
    try:
        pj = json.load(open(ppath))
    except Exception:
        pj = None
    rb["note"] = ("batches below 100 M samples (and RT_FUSE=2) run connect(r) + light(r) on a second stream beside round r + 1: the kernel times of "
                  "kernel_ms_per_step then overlap and do not add up to the step, and avg_launch_ms is the extend launches' own duration with that "
                  "company; batches of 100 M samples and more (the default workload) run one kernel at a time")
    # with N ranks the counter file is used when it was measured on this rank's share: the row shard renders H / N rows of every
    # frame, the counter file of the 1-GPU share ("--spp S/N": the same number of samples per rank) is the closest committed
    # stand-in and is NOT used -- frac stays null unless a file for exactly [workload, W, H, spp, world] exists
    khash = kernel_hash(build_info)
    if pj is not None and "by_world" in pj:  # one counter set per number of ranks (rank 0's rows of the N-rank shard, measured with --emulate-world N)
        pj = pj["by_world"].get(str(world))
    usable = (pj is not None and pj.get("kernel_hash") == khash and not args.qlearn and
              pj.get("workload") == [args.workload, W, H, spp] and pj.get("world", 1) == world)
    rb["pmc"] = {"file": "profiles/roofline_pmc.json", "used": bool(usable), "kernel_hash": khash,
                 "file_kernel_hash": pj.get("kernel_hash") if pj else None}
    if usable and sec > 0 and "ta_busy_avg" in pj["kernels"]["k_extend"]:
        k = pj["kernels"]["k_extend"]
        prof_sec = k["ms"] / k["launches"] * 1e-3  # the launch under the profiler (counters cost a few per cent)
        clock = k["clock_ghz"] * 1e9
        ta_cycles = k["ta_busy_avg"] * prof_sec * clock  # TA busy cycles per launch, avg over the TAs: work, not time
        ta_live = ta_cycles / (sec * clock)
        lane_ops = k["lane_ops_per_launch"]
        rb["achieved"] = round(ta_live, 4)
        rb["frac"] = round(ta_live, 4)
        rb["traffic"] = k["hbm_bytes_per_launch"]
        rb["sides"] = {
            "ta_busy": {"profiled": k["ta_busy_avg"], "profiled_max_over_tas": k["ta_busy_max"], "live": round(ta_live, 4), "cycles_per_vmem_inst": k.get("ta_cycles_per_vmem_inst"),
                        "vmem_insts_per_launch": int(k["vmem_insts"] / k["launches"]) if k.get("vmem_insts") else None},
            "l1_accesses_per_cu_cycle": k["l1_accesses_per_cu_cycle"],
            "valu_enabled_lane_frac_of_peak": round(lane_ops / sec / 78.6432e12, 4),  # 256 CUs x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 T lane-ops/s
            "valu": {"lanes_enabled": k["lanes_enabled"], "valu_pipe_busy": k["valu_pipe_busy"], "wave_wait_frac": k["wave_wait_frac"]},
            "hbm_counter_frac_of_peak": round(k["hbm_bytes_per_launch"] / sec / 8e12, 4),
            "hbm_counter_GBps": round(k["hbm_bytes_per_launch"] / sec / 1e9, 1),
            "hbm_algorithmic_demand_over_peak": rb["hbm"]["algorithmic_demand_over_hbm_peak"],
            "hbm_algorithmic_demand_note": "served by caches (L2 hit rate %.2f): > 1 is possible and says nothing about HBM" % k["l2_hit_rate"],
            "l2_hit_rate": k["l2_hit_rate"], "profile_launch_ms": round(k["ms"] / k["launches"], 4), "profile_clock_ghz": k["clock_ghz"]}
    return rb


def physical_cores():
    """(physical cores, logical CPUs) this process may run on, from /proc/cpuinfo's (physical id, core id) pairs."""
    allowed = os.sched_getaffinity(0)
    cores, cur = set(), {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif not line.strip():
                if "processor" in cur and int(cur["processor"]) in allowed:
                    cores.add((cur.get("physical id", "0"), cur.get("core id", cur["processor"])))
                cur = {}
    except Exception:
        pass
    return (len(cores) or len(allowed)), len(allowed)


def cpu_allowance():
    """CPUs' worth of time this process may use per second: the cgroup quota (cpu.max, v2; cfs_quota_us, v1), None when unlimited.
    A GPU box gives one GPU's job 16 of its 256 logical CPUs this way while the affinity mask still shows all of them: threads
    beyond the quota are throttled, not run (profiles/r05_cpu_scaling.txt: 128 threads reach 0.5-0.7x of what 16-32 do)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def frame_error(got, ref, floor=1e-3):
    """Relative error |got - ref| / max(|ref|, floor) over the finite entries (tests/conftest.py rel_err) and whether the
    non-finite ones agree by class."""
    import numpy as np
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    fa, fb = np.isfinite(got), np.isfinite(ref)
    cls_ok = bool(np.array_equal(fa, fb) and np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isposinf(got), np.isposinf(ref)))
    m = fa & fb
    err = np.abs(got[m] - ref[m]) / np.maximum(np.abs(ref[m]), floor)
    return (float(err.max()) if err.size else 0.0), cls_ok


def cpu_baseline(args, cfg, W, H, spp, gpu=None):
    """The oracle (CPU restatement, kind 'port') on the host cores: frames 0 .. F-1 of the same workload, OpenMP over
    scanlines like renderer.cpp:259, per-pixel RNG streams; F = the step's spp when that fits --cpu-seconds.  Threads =
    the physical cores this job may really use (the cgroup quota caps them: cpu_allowance).  value_1thread: one thread on
    every 8th scanline of one frame (a bounded sample of the same image: the GPU legs stay visible beside the CPU leg).
    gpu = (renderer, accumulator tensor, mode, checksum of the timed steps' frame): the SAME F frames are rendered once more
    on the device, outside the timed region, and compared with the oracle's accumulator as a whole -> (baseline, parity_check):
    the witness that the numbers above were measured on the reference's image (renderer.cpp:263-285), at full size."""
    import numpy as np
    from oracle import oracle_api as oa
    scenes = pkg("scenes")
    oa.build()
    phys, logical = physical_cores()
    allowed = cpu_allowance()
    threads = phys if allowed is None else max(1, min(phys, int(allowed + 0.5)))
    s = oa.OracleScene()
    scenes.REGISTRY[args.workload](s)
    s.set_raytracer(False)
    orr = oa.OracleRenderer(s, W, H)
    if "camera" in cfg:
        c = cfg["camera"]
        orr.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
    orr.clear()
    t0 = time.perf_counter()
    orr.render(0, 1, nthreads=threads)
    t1 = time.perf_counter() - t0
    frames = int(max(1, min(spp, args.cpu_seconds / max(t1, 1e-3))))
    if frames > 1:
        orr.render(1, frames - 1, nthreads=threads)
    dt = time.perf_counter() - t0
    ref = orr.accumulator()
    rows = list(range(4, H, 8))
    t0 = time.perf_counter()
    for y in rows:
        orr.render(0, 1, y0=y, y1=y + 1, nthreads=1)
    dt1 = time.perf_counter() - t0
    orr.close()
    s.close()
    base = {"value": round(W * H * frames / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "frames 0..%d of %dx%d of the same workload on %d threads (%.1f s)" % (frames - 1, W, H, threads, dt),
            "physical_cores": phys, "logical_cpus": logical, "cpu_allowance": allowed,
            "threads_note": "threads = min(physical cores, cgroup CPU quota): the box shows %d logical CPUs but this job may use %s of them" % (logical, "all" if allowed is None else "%.0f CPUs' worth of time" % allowed),
            "value_1thread": round(W * len(rows) / dt1 / 1e6, 4),
            "sample_1thread": "%d scanlines (every 8th) of one %dx%d frame on 1 thread (%.1f s)" % (len(rows), W, H, dt1)}
    if gpu is None:
        return base, None
    import torch
    r, acc, mode, timed_checksum = gpu
    acc.zero_()
    torch.cuda.synchronize()
    r.render(mode, 0, frames)
    r.synchronize()
    checksum = "%016x" % int(torch.sum(acc.view(torch.int32).to(torch.int64)).item() & 0xFFFFFFFFFFFFFFFF)
    got = acc.cpu().numpy()
    err, cls_ok = frame_error(got[..., :3], ref[..., :3])
    tol = 1e-4  # BASELINE.json north_star: accumulated radiance within 1e-4 relative
    parity = {"frames": frames, "pixels": W * H, "max_rel_err": err, "tolerance": tol, "nonfinite_class_equal": cls_ok,
              "frac_bit_identical": round(float((got[..., :3] == ref[..., :3]).mean()), 6),
              "same_frames_as_timed_step": frames == spp, "same_checksum_as_timed_step": (checksum == timed_checksum) if frames == spp else None,
              "oracle": "oracle/ (CPU restatement of renderer.cpp:128-236; parity unpinned: the reference holds no fixtures)",
              "ok": bool(cls_ok and err <= tol and (frames != spp or checksum == timed_checksum))}
    return base, parity


if __name__ == "__main__":
    main()
