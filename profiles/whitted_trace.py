"""One Whitted frame of the instanced glass / metal scene at 1080p, three times (for a rocprofv3 kernel trace)."""
import sys, importlib
sys.path.insert(0, ".")
ha = importlib.import_module("ray-and-pathtracer_amd.host_api"); scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
r = ha.HostRenderer(1920, 1080); d = scenes.REGISTRY["pretty_tlas"](r.scene, n_instances=8); r.commit()
c = d["camera"]; r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
for _ in range(3):
    r.clear(); r.render(ha.RT_MODE_WHITTED, 0, 1); r.synchronize()
r.close()
