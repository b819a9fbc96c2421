"""One line per scene: a path-mode Renderer::Tick at 1920x1080 without the host mirror (the part the kernels decide), and the
rays parked per Tick (rt_carry_stats, when the library has profiles/patches/carry.diff applied).  Usage (GPU box): [RT_CARRY=.. RT_CARRY_K=.. RT_FUSE=..] python profiles/tick_path.py"""
import sys, time, importlib, os
sys.path.insert(0, ".")
ha = importlib.import_module("ray-and-pathtracer_amd.host_api"); scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
out = []
for name, kw in (("mixed_small", {}), ("pretty_tlas", {"n_instances": 8})):
    r = ha.HostRenderer(1920, 1080); d = scenes.REGISTRY[name](r.scene, **kw); r.commit()
    if "camera" in d:
        c = d["camera"]; r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
    r.scene.set_raytracer(False)
    r.L.rth_renderer_set_download(r.h, 0)
    for _ in range(8):
        r.tick()
    stats = getattr(r, 'carry_stats', lambda: (0, 1))
    stats()
    n = 40
    t = time.perf_counter()
    for _ in range(n):
        r.tick()
    dt = (time.perf_counter() - t) / n
    parked, batches = stats()
    r.set_profiling(True); r.profile()
    for _ in range(5):
        r.tick()
    pr = r.profile(); r.set_profiling(False)
    out.append("%s %.3f ms (parked %.0f / Tick; %s)" % (name, dt * 1e3, parked / max(1, batches), " ".join("%s %.2f" % (k, v["ms"] / 5) for k, v in pr.items() if v["launches"])))
    r.close()
print("%-40s %s" % (" ".join("%s=%s" % (k, os.environ[k]) for k in ("RT_CARRY_K", "RT_FUSE", "RT_MIXED_MAX") if k in os.environ) or "(defaults)", " | ".join(out)), flush=True)
