"""Renderer::Tick per second (the reference's interactive loop: one frame per Tick, pixels resolved to the host every frame)
with and without refreshing the host mirror of the accumulator: the PCIe-inclusive rate of the drop-in host.
Usage (GPU box): python profiles/tick_time.py"""
import sys, time, importlib, ctypes as C
sys.path.insert(0, ".")
ha = importlib.import_module("ray-and-pathtracer_amd.host_api"); scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
for name, kw, w, h in (("mixed_small", {}, 1920, 1080), ("pretty_tlas", {"n_instances": 8}, 1920, 1080)):
    for path in (False, True):
        for download in (1, 0):
            r = ha.HostRenderer(w, h); d = scenes.REGISTRY[name](r.scene, **kw); r.commit()
            if "camera" in d:
                c = d["camera"]; r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
            r.scene.set_raytracer(not path)
            r.L.rth_renderer_set_download(r.h, download)
            for _ in range(6):
                r.tick()
            n = 20
            t = time.perf_counter()
            for _ in range(n):
                r.tick()
            dt = (time.perf_counter() - t) / n
            r.set_profiling(True); r.profile()
            for _ in range(5):
                r.tick()
            pr = r.profile(); r.set_profiling(False)
            kern = ", ".join("%s %.2f" % (k, v["ms"] / 5) for k, v in pr.items() if v["launches"])
            print("%s %dx%d %s Tick, accumulator mirrored to the host %s: %.2f ms per Tick (%.0f M primary samples/s); kernels per Tick (ms): %s" % (name, w, h, "path" if path else "Whitted", "every Tick" if download else "never", dt * 1e3, w * h / dt / 1e6, kern), flush=True)
            r.close()
