#!/bin/bash
# How many pair visits of the nearest-hit walk only dismiss a subtree (both children fail), and how many of those fail because of ray.t (measurement build
# _v/probe: the counts ride in the brute_tests / light_tests counters of the counting pass)
for w in config3 config4 config2; do ( cd _v/probe && timeout -k 10 300 python3 - $w <<'PY'
import sys, importlib
sys.path.insert(0, ".")
ha = importlib.import_module("ray-and-pathtracer_amd.host_api"); scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
name = sys.argv[1]
probe = ha.HostScene(); cfg = scenes.REGISTRY[name](probe); probe.close()
W, H, F = cfg["width"], cfg["height"], min(cfg["frames"], 8)
r = ha.HostRenderer(W, H); scenes.REGISTRY[name](r.scene); r.commit()
if "camera" in cfg:
    c = cfg["camera"]; r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
r.set_counting(ha.RT_COUNT_EXECUTED); r.counters()
r.clear(); r.render(ha.RT_MODE_PATH, 0, F)
near, occl = r.counters_split()
vis = near["inner_visits"] + near["tlas_inner"]
print("%s: %d frames: pair visits %d (BLAS %d + TLAS %d), both children fail %d (%.1f %%), of them because of ray.t %d (%.1f %% of all pair visits); prim tests %d, instance entries %d, rays %d" % (
    name, F, vis, near["inner_visits"], near["tlas_inner"], near["brute_tests"], 100.0 * near["brute_tests"] / vis, near["light_tests"], 100.0 * near["light_tests"] / vis, near["prim_tests"], near["instance_visits"], near["rays_nearest"]))
PY
); done
