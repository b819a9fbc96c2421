"""How long does ONE traversal step take when a lane walks alone (the drain of a launch, the tail of a Whitted pixel)?
Single-ray batches of rt_intersect_batch on the instanced glass / metal scene: steps from the counting launch (executed walk),
kernel time from the profiling events of the timed launch, for the rays of a coarse grid over the camera.  Also batches of 64
(one full wave) and 4096 rays for comparison.  Usage (GPU box): python profiles/lone_step.py"""
import sys, importlib
import numpy as np
sys.path.insert(0, ".")
ha = importlib.import_module("ray-and-pathtracer_amd.host_api"); scenes = importlib.import_module("ray-and-pathtracer_amd.scenes")
w, h = 1920, 1080
r = ha.HostRenderer(w, h); d = scenes.REGISTRY["pretty_tlas"](r.scene, n_instances=8); r.commit()
if "camera" in d:
    c = d["camera"]; r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
cam = r.camera()
pos, tl, tr, bl = (np.array(cam[k], dtype=np.float32) for k in range(4))
ys, xs = np.meshgrid(np.linspace(0.05, 0.95, 24), np.linspace(0.05, 0.95, 40), indexing="ij")
P = tl[None, :] + xs.reshape(-1, 1) * (tr - tl)[None, :] + ys.reshape(-1, 1) * (bl - tl)[None, :]
D = P - pos[None, :]; D /= np.linalg.norm(D, axis=1, keepdims=True)
O = np.repeat(pos[None, :], len(D), 0)
def steps_of(Ob, Db):
    r.set_counting(2); r.counters()
    r.find_nearest(Ob, Db)
    c = r.counters(); r.set_counting(0)
    return c["inner_visits"] + c["prim_tests"] + c["tlas_inner"] + 2 * c["instance_visits"]
def time_of(Ob, Db, reps=5):
    r.find_nearest(Ob, Db)
    r.set_profiling(True); r.profile()
    for _ in range(reps):
        r.find_nearest(Ob, Db)
    p = r.profile(); r.set_profiling(False)
    return sum(v["ms"] for v in p.values()) / reps * 1e3
st = np.array([steps_of(O[i:i + 1], D[i:i + 1]) for i in range(len(O))])
long = np.argsort(st)[-40:]
base = time_of(O[np.argmin(st):np.argmin(st) + 1], D[np.argmin(st):np.argmin(st) + 1])
print("shortest ray: %d steps, launch %.1f us" % (st.min(), base))
tot_s = tot_t = 0
for i in long:
    t = time_of(O[i:i + 1], D[i:i + 1])
    tot_s += st[i]; tot_t += t
    print("ray %4d: %4d steps, %.1f us -> %.0f ns per step (launch floor subtracted: %.0f)" % (i, st[i], t, t * 1e3 / st[i], (t - base) * 1e3 / max(1, st[i] - st.min())))
print("lone lane: %.0f ns per step over the 40 longest rays (floor subtracted: %.0f)" % (tot_t * 1e3 / tot_s, (tot_t - 40 * base) * 1e3 / (tot_s - 40 * st.min())))
for n in (64, 960):
    idx = np.argsort(st)[-n:]
    t = time_of(O[idx], D[idx]); s = steps_of(O[idx], D[idx])
    print("%d longest rays in one batch: %d steps, %.1f us, longest ray %d steps -> %.0f ns per step of the longest ray" % (n, s, t, st[idx].max(), (t - base) * 1e3 / st[idx].max()))
r.close()
