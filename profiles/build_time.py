import sys, importlib, time, numpy as np
sys.path.insert(0,".")
from oracle.oracle_api import OracleScene
ha = importlib.import_module("ray-and-pathtracer_amd.host_api"); scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
r = ha.HostRenderer(8,8)
for mesh in ("unity","BigB"):
    o = OracleScene()
    t0=time.perf_counter()
    (scenes.REGISTRY["pretty_tlas"](o, n_instances=2) if mesh=="unity" else scenes.REGISTRY["tlas_test2"](o, mesh=mesh))
    tv = o.mesh_tris(0)[0][:, :9]
    # host (C++ mirror) build time
    hs = ha.HostRenderer(8,8); t0=time.perf_counter(); (scenes.REGISTRY["pretty_tlas"](hs.scene, n_instances=2) if mesh=="unity" else scenes.REGISTRY["tlas_test2"](hs.scene, mesh=mesh)); t_host=time.perf_counter()-t0
    r.build_bvh(tv)
    ts=[]
    for _ in range(5):
        t0=time.perf_counter(); r.build_bvh(tv); ts.append(time.perf_counter()-t0)
    print(mesh, len(tv), "tris: device build %.2f ms (min of 5, incl. upload/readback/renumber); host scene construction incl. load+build(s) %.1f ms"%(min(ts)*1e3, t_host*1e3))
    o2 = OracleScene(); m = o2.mesh_raw(1, o2.diffuse(0.8,(1,1,1)), tv)
    t0=time.perf_counter(); o2.build(0); print("   CPU restatement of bvh::Build (one core): %.2f ms"%((time.perf_counter()-t0)*1e3))
