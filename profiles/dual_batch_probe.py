"""Would two half-batches on two streams beat one batch?  Probe with what exists: TWO contexts on one GPU, each renders half of the
step's frames of config 3 from its own host thread (ctypes releases the GIL), so that one half's extend runs beside the other's
shade.  The two accumulators are NOT summed here (a timing probe only).  Usage (GPU box): python profiles/dual_batch_probe.py [spp]"""
import sys, time, threading, importlib
sys.path.insert(0, ".")
ha = importlib.import_module("ray-and-pathtracer_amd.host_api"); scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
delay = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
probe = ha.HostScene(); cfg = scenes.REGISTRY["config3"](probe); probe.close()
W, H = cfg["width"], cfg["height"]
def make():
    r = ha.HostRenderer(W, H); scenes.REGISTRY["config3"](r.scene); r.commit()
    if "camera" in cfg:
        c = cfg["camera"]; r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
    return r
a, b = make(), make()
def one(n=5):
    a.clear(); a.render(ha.RT_MODE_PATH, 0, spp); a.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        a.clear(); a.render(ha.RT_MODE_PATH, 0, spp)
    a.synchronize()
    return (time.perf_counter() - t) / n
def two(n=5, parts=2):
    per = spp // parts
    def work(r, f0s, lag):
        if lag: time.sleep(lag)
        for f0 in f0s:
            r.render(ha.RT_MODE_PATH, f0, per)
    def step():
        a.clear(); b.clear(); a.synchronize(); b.synchronize()
        ta = threading.Thread(target=work, args=(a, list(range(0, spp, 2 * per)), 0.0)); tb = threading.Thread(target=work, args=(b, list(range(per, spp, 2 * per)), delay))
        ta.start(); tb.start(); ta.join(); tb.join()
        a.synchronize(); b.synchronize()
    step()
    t = time.perf_counter()
    for _ in range(n):
        step()
    return (time.perf_counter() - t) / n
print("config 3, %d spp: one context, one batch %.3f ms" % (spp, one() * 1e3), flush=True)
for parts in (2, 4, 8):
    print("two contexts, %d batches of %d frames alternating (second thread starts %.1f ms late): %.3f ms" % (parts, spp // parts, delay * 1e3, two(parts=parts) * 1e3), flush=True)
print("one context again %.3f ms" % (one() * 1e3))
a.close(); b.close()
