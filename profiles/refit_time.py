#!/usr/bin/env python3
"""Time Scene::SetTime on the device (rt_set_time: k_animate + bvh::Refit level by level, csrc/rt_api.hip) for a scene
BVH over BigB.obj (11,830 triangles) and over the 51,200-triangle tower, against the oracle's CPU SetTime + Refit.
Usage (GPU box): python profiles/refit_time.py"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ha = importlib.import_module("ray-and-pathtracer_amd.host_api")
scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
assets = importlib.import_module("ray-and-pathtracer_amd.assets")
from oracle import oracle_api as oa


def bigb_scene(b):
    m = b.diffuse(0.8, (1, 0, 0), 0.0, 1, 1)
    b.mesh_obj(1, assets.obj_path("BigB"), m, (0, 0.5, 0), 1)
    b.plane(0, b.diffuse(0.8, (1, 1, 1), 0.0, 1.0, 4), (0, 1, 0), 0)
    b.build(0)


out = {}
for name, fn in (("BigB.obj scene BVH (11,830 triangles)", bigb_scene), ("tower (51,200 triangles)", scenes.tower_scene)):
    r = ha.HostRenderer(64, 64)
    fn(r.scene)
    r.commit()
    r.set_time(0.1); r.synchronize()
    n = 50
    t0 = time.perf_counter()
    for i in range(n):
        r.set_time(0.1 + 0.01 * i)
    r.synchronize()
    gpu_us = (time.perf_counter() - t0) / n * 1e6
    o = oa.OracleScene()
    fn(o)
    t0 = time.perf_counter()
    for i in range(5):
        o.set_time(0.1 + 0.01 * i)
    cpu_us = (time.perf_counter() - t0) / 5 * 1e6
    out[name] = {"rt_set_time_us": round(gpu_us, 1), "oracle_cpu_set_time_us": round(cpu_us, 1)}
    r.close()
print(json.dumps(out, indent=1))
