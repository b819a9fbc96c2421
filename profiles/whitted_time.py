"""Whitted frames (Renderer::Trace) per second on a few scenes: informational, not a bench line."""
import sys, time, importlib
sys.path.insert(0, ".")
ha = importlib.import_module("ray-and-pathtracer_amd.host_api"); scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
for name, kw, w, h in (("background", {}, 1280, 720), ("mixed_small", {}, 1920, 1080), ("pretty_tlas", {"n_instances": 8}, 1920, 1080), ("scene3", {"force_diffuse": False}, 1920, 1080)):
    r = ha.HostRenderer(w, h); d = scenes.REGISTRY[name](r.scene, **kw); r.commit()
    if "camera" in d:
        c = d["camera"]; r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
    r.render(ha.RT_MODE_WHITTED, 0, 1); r.synchronize()
    ts = []
    for _ in range(5):
        t = time.perf_counter(); r.render(ha.RT_MODE_WHITTED, 0, 1); r.synchronize(); ts.append(time.perf_counter() - t)
    print("%s %dx%d Whitted: %.2f ms per frame (%.0f Mrays/s primary)" % (name, w, h, min(ts) * 1e3, w * h / min(ts) / 1e6), flush=True)
    r.close()
