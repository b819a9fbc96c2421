#!/usr/bin/env python3
"""Generate tests/golden/oracle_vectors.npz from the oracle (oracle/).

The reference has no tests, fixtures or golden images and cannot be built in this
image without stand-ins for Windows/MSVC headers, so these vectors do NOT come from the reference:
they pin the oracle against drift (any edit that changes its arithmetic fails the CPU test tier) and
give the GPU tier fixed expected values that do not depend on rebuilding the oracle.
Run:  python tests/golden/make_golden.py
"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_api as oa
scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
import ctypes as C


def main():
    out = {}
    L = oa.lib()
    # RNG stream and hemisphere sampling
    for seed in (0, 1, 0x12345678, 0xFFFFFFFE):
        u = np.zeros(64, np.uint32); f = np.zeros(64, np.float32)
        L.orc_rng_stream(C.c_uint(seed), 64, u.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p))
        out["rng_u_%x" % seed] = u; out["rng_f_%x" % seed] = f
    rng = np.random.default_rng(1)
    nrm = rng.normal(size=(64, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True).astype(np.float32)
    hemi = np.zeros((64, 3), np.float32)
    L.orc_hemisphere(C.c_uint(99), 64, nrm.ctypes.data_as(C.c_void_p), hemi.ctypes.data_as(C.c_void_p))
    out["hemi_normals"] = nrm; out["hemi_out"] = hemi
    # fresnel / refract
    I = rng.normal(size=(128, 3)).astype(np.float32); I /= np.linalg.norm(I, axis=1, keepdims=True).astype(np.float32)
    N = rng.normal(size=(128, 3)).astype(np.float32); N /= np.linalg.norm(N, axis=1, keepdims=True).astype(np.float32)
    kr = np.zeros(128, np.float32); rf = np.zeros((128, 3), np.float32)
    L.orc_fresnel(128, I.ctypes.data_as(C.c_void_p), N.ctypes.data_as(C.c_void_p), C.c_float(1.5), kr.ctypes.data_as(C.c_void_p))
    L.orc_refract(128, I.ctypes.data_as(C.c_void_p), N.ctypes.data_as(C.c_void_p), C.c_float(1 / 1.5), rf.ctypes.data_as(C.c_void_p))
    out["fr_I"] = I; out["fr_N"] = N; out["fr_kr"] = kr; out["fr_refract"] = rf
    # per scene: BVH dump digest, primary hit maps, whitted + path accumulators at 48x32
    for name, kw in (("background", {}), ("mixed_small", {}), ("scene3", {"force_diffuse": False}), ("tlas_test2", {})):
        s = oa.OracleScene()
        d = scenes.REGISTRY[name](s, **kw)
        key = name
        if not d["tlas"]:
            b = s.bvh_dump(-1)
            out[key + "_nodes"] = np.delete(b["nodes"], 1, axis=0); out[key + "_prim_idx"] = b["prim_idx"]
        else:
            b = s.bvh_dump(0)
            out[key + "_nodes"] = np.delete(b["nodes"], 1, axis=0); out[key + "_prim_idx"] = b["prim_idx"]
            out[key + "_tlas"] = s.tlas_dump()
        r = oa.OracleRenderer(s, 48, 32)
        obj, t, cnt = r.primary_hits(1e-6)
        out[key + "_obj"] = obj; out[key + "_t"] = t
        out[key + "_cnt"] = np.array([cnt[k] for k in oa.COUNTER_NAMES], dtype=np.uint64)
        s.set_raytracer(True); r.clear(); r.render(0, 1)
        out[key + "_whitted"] = r.accumulator()
        s.set_raytracer(False); r.clear(); r.render(0, 4)
        out[key + "_path4"] = r.accumulator()
        r.close(); s.close()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
