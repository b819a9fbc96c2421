"""Why are 128 threads of the oracle only ~11x one thread?  The box's CPU allowance (cgroup quota, affinity) and the
oracle's frame rate at 1..128 threads with OpenMP's default placement and with OMP_PROC_BIND=spread OMP_PLACES=cores.
Usage (GPU box): python profiles/cpu_scaling.py"""
import os, sys, time, subprocess, importlib
sys.path.insert(0, ".")
def show(path):
    try:
        print("%s: %s" % (path, open(path).read().strip()))
    except Exception as e:
        print("%s: (%s)" % (path, type(e).__name__))
if len(sys.argv) == 1:
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpu.stat"):
        show(p)
    print("affinity: %d cpus; os.cpu_count %d; loadavg %s" % (len(os.sched_getaffinity(0)), os.cpu_count(), open("/proc/loadavg").read().strip()))
    for env in ({}, {"OMP_PROC_BIND": "spread", "OMP_PLACES": "cores"}):
        for n in (1, 4, 8, 16, 32, 64, 128):
            e = dict(os.environ); e.update(env)
            out = subprocess.run([sys.executable, __file__, str(n)], env=e, capture_output=True, text=True).stdout.strip()
            print("%-40s threads %3d: %s" % (" ".join("%s=%s" % kv for kv in env.items()) or "(default placement)", n, out), flush=True)
    show("/sys/fs/cgroup/cpu.stat")
else:
    n = int(sys.argv[1])
    from oracle import oracle_api as oa
    scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
    oa.build()
    s = oa.OracleScene(); cfg = scenes.REGISTRY["config3"](s); s.set_raytracer(False)
    W, H = 1920, 1080
    orr = oa.OracleRenderer(s, W, H)
    if "camera" in cfg:
        c = cfg["camera"]; orr.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
    rows = H if n >= 8 else H // 8
    orr.render(0, 1, y0=0, y1=rows, nthreads=n)
    t = time.perf_counter(); f = 2 if n >= 8 else 1
    orr.render(1, f, y0=0, y1=rows, nthreads=n)
    dt = time.perf_counter() - t
    print("%.3f Mrays/s (%d rows x %d frames in %.2f s)" % (W * rows * f / dt / 1e6, rows, f, dt))
