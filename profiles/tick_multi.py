"""rapt::Renderer::Tick with one context against two and three contexts on the SAME GPU (devices [0], [0, 0], [0, 0, 0]): what the
multi-context plumbing itself costs per Tick -- worker hand-off, the push gather, the stream waits -- when the work is the same
(the contexts share the device, so nothing gets faster).  Usage (GPU box): python profiles/tick_multi.py"""
import sys, time, importlib
sys.path.insert(0, ".")
ha = importlib.import_module("ray-and-pathtracer_amd.host_api"); scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
w, h = 1920, 1080
for path in (False, True):
    for devs in (None, [0, 0], [0, 0, 0]):
        r = ha.HostRenderer(w, h, 0, devices=devs); d = scenes.REGISTRY["pretty_tlas"](r.scene, n_instances=8); r.commit()
        c = d["camera"]; r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
        r.scene.set_raytracer(not path)
        r.L.rth_renderer_set_download(r.h, 0)
        for _ in range(8):
            r.tick()
        n = 30
        t = time.perf_counter()
        for _ in range(n):
            r.tick()
        dt = (time.perf_counter() - t) / n
        print("pretty_tlas %dx%d %s Tick, contexts %s: %.3f ms per Tick" % (w, h, "path" if path else "Whitted", devs or [0], dt * 1e3), flush=True)
        r.close()
