"""Manual helper (not collected by pytest): time one rank's share of the bench frame for N = 1, 2, 4, 8
on a single GPU (rows r, r+N, ... of the 1080p x 64 spp frame) to estimate strong-scaling efficiency."""
import sys, time, importlib
sys.path.insert(0, ".")
ha = importlib.import_module("ray-and-pathtracer_amd.host_api"); scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
w, h, spp = 1920, 1080, 64
r = ha.HostRenderer(w, h); d = scenes.config3(r.scene); r.commit()
c = d["camera"]; r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
base = None
for n in (1, 2, 4, 8):
    cnt = len(range(0, h, n))
    for rep in range(3):
        r.clear(); r.synchronize()
        t = time.perf_counter(); r.render_rows(ha.RT_MODE_PATH, 0, spp, 0, n, cnt); r.synchronize(); dt = time.perf_counter() - t
    base = base or dt
    r.set_profiling(True); r.profile()
    r.clear(); r.render_rows(ha.RT_MODE_PATH, 0, spp, 0, n, cnt); r.synchronize()
    pr = r.profile(); r.set_profiling(False)
    print("   kernels:", {k: (v["launches"], round(v["ms"], 2)) for k, v in pr.items() if v["launches"]}, flush=True)
    print("N=%d rank0 share: %.2f ms  -> speedup %.2fx (ideal %d)" % (n, dt * 1e3, base / dt, n), flush=True)
