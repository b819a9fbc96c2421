"""Manual stress helper (not collected by pytest): time rt_render for growing frame counts."""
import sys, time, importlib
sys.path.insert(0, ".")
ha = importlib.import_module("ray-and-pathtracer_amd.host_api"); scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
r = ha.HostRenderer(w, h); d = scenes.REGISTRY[name](r.scene); r.commit()
if "camera" in d:
    c = d["camera"]; r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
for n in [int(x) for x in sys.argv[4:]]:
    r.clear(); r.synchronize()
    t = time.perf_counter(); r.render(ha.RT_MODE_PATH, 0, n); r.synchronize(); dt = time.perf_counter() - t
    print(name, w, h, "frames", n, "%.3f s" % dt, "%.1f Mrays/s" % (w * h * n / dt / 1e6), flush=True)
