import os, sys, importlib
import numpy as np
sys.path.insert(0, "/root/repo")
ha = importlib.import_module("ray-and-pathtracer_amd.host_api")
scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
from oracle import oracle_api as oa
w, h = 160, 90
res = {}
for key, env in (("slot", {"RT_STREAM": "0"}), ("stream", {"RT_STREAM": "1"}), ("stream_nd", {"RT_STREAM": "1", "RT_DECIDE": "0"}), ("stream_serial", {"RT_STREAM": "1", "RT_FUSE": "0"})):
    for k in ("RT_STREAM", "RT_DECIDE", "RT_FUSE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    r = ha.HostRenderer(w, h)
    d = scenes.pretty_tlas(r.scene, n_instances=4)
    r.commit()
    c = d["camera"]
    r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
    o = oa.OracleScene(); scenes.pretty_tlas(o, n_instances=4)
    orr = oa.OracleRenderer(o, w, h); orr.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
    pO, pD = orr.primary_rays()
    pO, pD = pO[::7].copy(), pD[::7].copy()
    res[key] = [r.trace_batch(ha.RT_MODE_PATH, pO, pD, depth, 99) for depth in (0, 1, 2, 4)]
    res[key] += [r.trace_batch(ha.RT_MODE_PATH, pO, pD, depth, 99) for depth in (0, 1, 2, 4)]
    r.close()
for key in res:
    for i, (a, b) in enumerate(zip(res["slot"], res[key])):
        bad = np.nonzero((a.view(np.uint32) != b.view(np.uint32)).any(-1))[0]
        print(key, "call", i, "mismatches", len(bad), bad[:8], a[bad[:3]], b[bad[:3]])
