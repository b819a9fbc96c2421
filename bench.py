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
reported beside it.  roofline: dominant kernel = k_extend (Scene::FindNearest); achieved =
algorithmic bytes of its launches (SURVEY.md 8d formula over the kernel's own work counters,
gathered in an untimed counting pass of the identical, deterministic workload) / its device time
measured with HIP events on its stream during the timed steps.  cpu_baseline: the oracle
(oracle/, the CPU restatement; kind "port") on this box's host cores, on a bounded sample (whole
1080p frames of the same workload, as many as fit ~15 s).
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
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 through torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; this benchmark has no CPU path")
    # RAPT_DIST_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: ranks share
    # devices (local_rank % device_count) and the gather goes through host memory.  The driver's
    # multi-GPU runs use the default: one GPU per rank, "nccl" (= RCCL over xGMI).
    backend = os.environ.get("RAPT_DIST_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    ha, scenes, dpar = pkg("host_api"), pkg("scenes"), pkg("distributed")
    # one rank (re)builds stale libraries; the others wait, they would race in the same directory
    if rank == 0:
        ha.build()
    if world > 1:
        dist.barrier()

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
    row_first, row_stride, row_count = dpar.shard_rows(H, rank, world)

    def step():
        acc.zero_()
        torch.cuda.synchronize()
        r.render_rows(mode, 0, spp, row_first, row_stride, row_count)
        r.synchronize()
        if backend == "nccl" or world == 1:
            dpar.gather_rows(acc, rank, world, 0)
        else:
            host = acc.cpu()
            dpar.gather_rows(host, rank, world, 0)
            if rank == 0:
                acc.copy_(host)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        r.synchronize()

    # ---- untimed: counting pass (algorithmic work of this rank's rows), then warmup ----
    r.set_counting(ha.RT_COUNT_EXECUTED)  # the walk the timed kernels make (same results, fewer TLAS / instance visits than the reference's)
    r.counters()
    step()
    near, occl = r.counters_split()
    r.set_counting(False)
    for _ in range(args.warmup):
        step()

    # ---- timed region: exactly K steps ----
    r.set_profiling(True)
    r.profile()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = r.profile()
    r.set_profiling(False)

    red_dev = "cuda" if backend == "nccl" else "cpu"
    t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    cnt = torch.tensor([near["rays_nearest"], occl["rays_occluded"]], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    dt = float(t.item())
    rays_all = float(cnt.sum().item())  # per step, all ranks

    if rank == 0:
        sec_per_step = dt / args.steps
        value = W * H * spp / sec_per_step / 1e6
        ext = prof["extend"]
        bytes_extend = ha.algorithmic_bytes(near, executed=True)  # this rank, one step
        launches_per_step = ext["launches"] / args.steps
        avg_ms = ext["ms"] / max(1, ext["launches"])
        achieved = (bytes_extend / max(1.0, launches_per_step)) / (avg_ms * 1e-3) / 1e9 if ext["launches"] else 0.0
        peak = 8000.0  # GB/s, MI355X HBM3E peak (MI355X_MICROARCH.md)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == args.workload and tj.get("width") == W and tj.get("height") == H and tj.get("spp") == spp:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s at 1920×1080×64spp",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(sec_per_step * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s, %dx%d, %d spp, path integrator" % (args.workload, cfg["name"], W, H, spp),
                       "parallelism": "row-interleaved pixel shard x%d + accumulator gather to rank 0" % world,
                       "rays_definition": "value counts primary pixel samples (reference's Mrays/s, renderer.cpp:300); all_rays counts every FindNearest + IsOccluded query"},
            "all_rays_mrays_per_s": round(rays_all / sec_per_step / 1e6, 3),
            "rays_per_step": {"nearest": int(cnt[0].item()), "occluded": int(cnt[1].item())},
            "roofline": {"bound": "hbm", "kernel": "k_extend (Scene::FindNearest)", "achieved": round(achieved, 2), "peak": peak, "unit": "GB/s",
                         "frac": round(achieved / peak, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(bytes_extend / max(1.0, launches_per_step)),
                         "algorithmic_work_per_step": {k: int(near[k]) for k in ("inner_visits", "prim_tests", "tlas_inner", "instance_visits", "brute_tests", "rays_nearest")},
                         "avg_launch_ms": round(avg_ms, 5), "launches_per_step": launches_per_step,
                         "kernel_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in prof.items() if v["launches"]}},
        }
        out["frame_checksum"] = "%016x" % int(torch.sum(acc.view(torch.int32).to(torch.int64)).item() & 0xFFFFFFFFFFFFFFFF)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, cfg, W, H)
        print(json.dumps(out), flush=True)
    r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, cfg, W, H):
    """The oracle (CPU restatement, kind 'port') on the host cores: whole frames of the same
    workload, OpenMP over scanlines like renderer.cpp:259, per-pixel RNG streams."""
    from oracle import oracle_api as oa
    scenes = pkg("scenes")
    oa.build()
    cores = len(os.sched_getaffinity(0))
    s = oa.OracleScene()
    scenes.REGISTRY[args.workload](s)
    s.set_raytracer(False)
    orr = oa.OracleRenderer(s, W, H)
    if "camera" in cfg:
        c = cfg["camera"]
        orr.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
    t0 = time.perf_counter()
    orr.render(0, 1, nthreads=cores)
    t1 = time.perf_counter() - t0
    frames = int(max(1, min(64, args.cpu_seconds / max(t1, 1e-3))))
    t0 = time.perf_counter()
    orr.render(1, frames, nthreads=cores)
    dt = time.perf_counter() - t0
    orr.close()
    s.close()
    return {"value": round(W * H * frames / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d frame(s) of %dx%d of the same workload (%.1f s)" % (frames, W, H, dt)}


if __name__ == "__main__":
    main()
