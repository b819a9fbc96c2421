import sys, importlib
import numpy as np
sys.path.insert(0, ".")
ha = importlib.import_module("ray-and-pathtracer_amd.host_api"); scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
probe = ha.HostScene(); cfg = scenes.REGISTRY["config5"](probe); probe.close()
W, H = cfg["width"], cfg["height"]
r = ha.HostRenderer(W, H); scenes.REGISTRY["config5"](r.scene); r.commit()
if "camera" in cfg:
    c = cfg["camera"]; r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
qbox = cfg.get("qbox", ((-12.0, -2.0, -8.0), (12.0, 10.0, 16.0)))
for mask in (3, 0):
    r.qlearn_enable(16, qbox[0], qbox[1], 0.3, 0.2, 1.0, mask)
    r.clear(); r.render_rows(ha.RT_MODE_PATH, 0, 32, 0, 1, H); r.synchronize()
    sums, cnts = r.qlearn_sums()
    print("mask", mask, "frames 32: rewards", int(cnts.sum()), "keys used", int((cnts > 0).sum()), "of", cnts.size, "max per key", int(cnts.max()), "mean per used key %.0f" % (cnts.sum() / max(1, (cnts > 0).sum())), "max sum / 2^16 = %.0f" % (sums.max() / 65536.0), flush=True)
r.close()
