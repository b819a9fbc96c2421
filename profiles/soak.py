"""Soak: the same frames again and again must be the same bits (a race in a queue, a work head or a position scan would show as a
different accumulator now and then).  Configs 2-4 at full size in path mode, a Whitted frame and a Q-learning batch pair each,
for about SECONDS seconds per scene.  Usage (GPU box): python profiles/soak.py [seconds_per_scene]"""
import sys, time, importlib, hashlib
import numpy as np
sys.path.insert(0, ".")
ha = importlib.import_module("ray-and-pathtracer_amd.host_api"); scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8)).hexdigest()[:16]
bad = 0
for name in ("config2", "config3", "config4"):
    probe = ha.HostScene(); cfg = scenes.REGISTRY[name](probe); probe.close()
    W, H, F = cfg["width"], cfg["height"], min(cfg["frames"], 64)
    r = ha.HostRenderer(W, H); scenes.REGISTRY[name](r.scene); r.commit()
    if "camera" in cfg:
        c = cfg["camera"]; r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
    ref = {}
    t0, n = time.time(), 0
    while time.time() - t0 < secs:
        for what in ("path", "path_rows", "whitted", "qlearn"):
            r.clear()
            if what == "path":
                r.render(ha.RT_MODE_PATH, 0, F)
            elif what == "path_rows":  # three interleaved shards, like three ranks
                for k in range(3):
                    r.render_rows(ha.RT_MODE_PATH, 0, F, k, 3, len(range(k, H, 3)))
            elif what == "whitted":
                r.scene.set_raytracer(True); r.render(ha.RT_MODE_WHITTED, 0, 1); r.scene.set_raytracer(False)
            else:
                r.qlearn_enable(8, (-12, -2, -8), (12, 10, 16), 0.3, 0.2, 1.0, 3)
                r.render(ha.RT_MODE_PATH, 0, 8); r.qlearn_apply(); r.render(ha.RT_MODE_PATH, 8, 8)
                tab = r.qlearn_table(); r.qlearn_disable()
            d = digest(r.accumulator()) + ("" if what != "qlearn" else digest(tab))
            if what == "path_rows":
                what_key = "path"  # the shards must give the unsharded frame
            else:
                what_key = what
            if what_key in ref and ref[what_key] != d:
                bad += 1
                print("MISMATCH", name, what, ref[what_key], d, flush=True)
            ref.setdefault(what_key, d)
        n += 1
    print("%s %dx%d x %d frames: %d rounds of (path, path in three shards, whitted, qlearn pair) in %.0f s, digests %s" % (name, W, H, F, n, time.time() - t0, ref), flush=True)
    r.close()
print("soak: %d mismatches" % bad)
sys.exit(1 if bad else 0)
