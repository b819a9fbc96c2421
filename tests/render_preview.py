import sys, importlib
sys.path.insert(0,'.')
import numpy as np
from PIL import Image
ha = importlib.import_module("ray-and-pathtracer_amd.host_api"); scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
for name, w, h, spp in (("config3", 640, 360, 32), ("config5", 640, 360, 16), ("config4", 640, 360, 16), ("config2", 640, 360, 16)):
    r = ha.HostRenderer(w, h); d = scenes.REGISTRY[name](r.scene); r.commit()
    if "camera" in d:
        c = d["camera"]; r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
    r.render(ha.RT_MODE_PATH, 0, spp)
    px = r.resolve(spp)
    img = np.stack([(px>>16)&255, (px>>8)&255, px&255], -1).astype(np.uint8)
    Image.fromarray(img).save("gpurun_out/preview_%s.png" % name)
    r.close()
print("ok")
