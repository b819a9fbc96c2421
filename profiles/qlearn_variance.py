"""Does guiding pay?  Per-ray variance of Sample() over seeds, guided (after learning on a larger frame) against the uniform-hemisphere
sampler, on the probe scene (sky + directional lights: every path finite).  Usage (GPU box): python profiles/qlearn_variance.py [train_w train_h batches]"""
import sys, importlib
import numpy as np
sys.path.insert(0, ".")
ha = importlib.import_module("ray-and-pathtracer_amd.host_api"); scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
from oracle import oracle_api as oa
tw, th, nb = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (320, 200, 10)
box = ((-4, -1, -4), (4, 5, 6))
o = oa.OracleScene(); d = scenes.REGISTRY["qlearn_probe"](o)
orr = oa.OracleRenderer(o, 64, 40)
if "camera" in d:
    c = d["camera"]; orr.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
pO, pD = orr.primary_rays()
r = ha.HostRenderer(tw, th); scenes.REGISTRY["qlearn_probe"](r.scene); r.commit()
if "camera" in d:
    r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
def lum(sb):
    v = r.trace_batch(ha.RT_MODE_PATH, pO, pD, 4, sb).astype(np.float64)
    return 0.2126 * v[:, 0] + 0.7152 * v[:, 1] + 0.0722 * v[:, 2]
seeds = [0x5EED0000 + 7919 * k for k in range(32)]
u = np.stack([lum(s) for s in seeds])
print("uniform: mean %.5f  per-ray variance %.5f" % (u.mean(), u.var(axis=0, ddof=1).mean()), flush=True)
for eps in (0.2, 0.5):
    r.qlearn_enable(8, box[0], box[1], 0.3, eps, 1.0, 0)
    for b in range(nb):
        r.render(ha.RT_MODE_PATH, 4 * b, 4); r.qlearn_apply()
        if b in (0, 2, nb - 1):
            g = np.stack([lum(s) for s in seeds])
            print("eps %.1f after %2d batches of 4 frames at %dx%d: mean %.5f  per-ray variance %.5f  ratio to uniform %.3f" % (eps, b + 1, tw, th, g.mean(), g.var(axis=0, ddof=1).mean(), g.var(axis=0, ddof=1).mean() / u.var(axis=0, ddof=1).mean()), flush=True)
    r.qlearn_disable()
r.close()
