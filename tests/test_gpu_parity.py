"""Parity of the HIP path (through the C ABI of include/rt_amd.h) with the oracle, on the GPU.

Bar: hit ids (objIdx), material indices and occlusion flags bit-exact; t and normals bit-exact
(same arithmetic, no contraction); accumulated radiance within 1e-4 relative (BASELINE.json
north_star), non-finite pixels compared by class (a directly viewed light is +inf in
the reference)."""
import os

import numpy as np
import pytest

from conftest import random_rays, rel_err

pytestmark = pytest.mark.gpu

RADIANCE_TOL = 1e-4  # relative, north_star


def make_pair(scene_fn, oracle_api, host_api, w, h, **kw):
    o = oracle_api.OracleScene()
    d = scene_fn(o, **kw)
    r = host_api.HostRenderer(w, h)
    scene_fn(r.scene, **kw)
    r.commit()
    orr = oracle_api.OracleRenderer(o, w, h)
    if "camera" in d:
        c = d["camera"]
        orr.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
        r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
    return o, orr, r, d


SCENES = [
    ("background", {}),
    ("mixed_small", {}),
    ("mixed_small", {"split": 1}),
    ("mixed_small", {"split": 2}),
    ("mixed_small", {"split": 3}),
    ("scene3", {"force_diffuse": False}),
    ("tlas_test2", {}),
    ("tlas_test2", {"mesh": "BigB"}),
]


@pytest.mark.parametrize("name,kw", SCENES)
def test_find_nearest_and_occlusion(name, kw, scenes, oracle_api, host_api):
    o, orr, r, d = make_pair(scenes.REGISTRY[name], oracle_api, host_api, 64, 48, **kw)
    O, D = random_rays(6000, 42)
    pO, pD = orr.primary_rays()
    O, D = np.concatenate([O, pO]), np.concatenate([D, pD])
    for t_min in (1e-6, 0.001):
        ref = o.find_nearest(O, D, t_min=t_min)
        r.set_counting(True)
        r.counters()
        got = r.find_nearest(O, D, t_min=t_min)
        cnt = r.counters()
        r.set_counting(False)
        assert np.array_equal(got["obj"], ref["obj"])
        assert np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))
        hit = ref["obj"] != -1
        assert np.array_equal(got["mat"][hit & (ref["mat"] >= 0)], ref["mat"][hit & (ref["mat"] >= 0)])
        assert np.array_equal(got["normal"][hit].view(np.uint32), ref["normal"][hit].view(np.uint32))
        for k in ("inner_visits", "prim_tests", "tlas_inner", "instance_visits", "rays_nearest", "brute_tests", "light_tests"):
            assert cnt[k] == ref["counters"][k], k
    # shadow-style queries: bounded tmax
    tmax = np.random.default_rng(3).uniform(0.1, 12.0, len(O)).astype(np.float32)
    ref = o.is_occluded(O, D, tmax)
    r.set_counting(True)
    r.counters()
    got = r.is_occluded(O, D, tmax)
    cnt = r.counters()
    assert np.array_equal(got, ref["occluded"])
    for k in ("inner_visits", "prim_tests", "tlas_inner", "instance_visits", "rays_occluded"):
        assert cnt[k] == ref["counters"][k], k
    # the timed kernels (counting off): the 4-wide walk for clean rays, the binary walk for the rest (the axis-aligned
    # directions among these rays are not clean) -- same flags
    r.set_counting(False)
    assert np.array_equal(r.is_occluded(O, D, tmax), ref["occluded"])
    assert np.array_equal(r.is_occluded(O, D), o.is_occluded(O, D)["occluded"])
    # tmax on the nearest-hit query too
    ref = o.find_nearest(O, D, tmax=tmax, t_min=1e-6)
    got = r.find_nearest(O, D, tmax=tmax, t_min=1e-6)
    assert np.array_equal(got["obj"], ref["obj"])
    assert np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))
    r.close()


@pytest.mark.parametrize("name,kw", [("mixed_small", {}), ("scene3", {"force_diffuse": False}), ("tlas_test2", {"mesh": "BigB"}), ("pretty_tlas", {"n_instances": 8})])
def test_wide_walk_equals_binary_walk(name, kw, scenes, oracle_api, host_api, monkeypatch):
    """the 4-wide nodes (csrc/rt_scene_dev.h wide[], built at upload by collapsing the reference's
    binary tree; the reference's own 4-wide variant is bvh.cpp:335-512, :658-761).  Scene::IsOccluded through them
    (RT_WIDE=1; off by default because it measured slower) must give the oracle's flags for every ray -- clean rays
    take the wide walk, the others (axis-aligned directions here) are handed back to the binary walk -- and a path
    frame whose shadow rays took it must equal the default frame bit for bit."""
    frames = {}
    for wide in ("0", "1", "8"):
        # "8": the 8-wide nodes with quantised child boxes (wide8[] / leafBox[], RT_WIDE8=1): inner boxes rounded outwards, the
        # leaf's own exact box tested before its primitives -- the same flags for every ray, the same frames
        monkeypatch.setenv("RT_WIDE", "0" if wide == "8" else wide)
        monkeypatch.setenv("RT_WIDE8", "1" if wide == "8" else "0")
        o, orr, r, d = make_pair(scenes.REGISTRY[name], oracle_api, host_api, 96, 54, **kw)
        if wide != "0":
            O, D = random_rays(40000, 9, center=(0.0, 1.0, 3.0), spread=6.0)
            pO, pD = orr.primary_rays()
            O, D = np.concatenate([O, pO]), np.concatenate([D, pD])
            rng = np.random.default_rng(5)
            for tmax in (None, rng.uniform(0.1, 12.0, len(O)).astype(np.float32), np.full(len(O), 2.5, np.float32)):
                ref = o.is_occluded(O, D, tmax)["occluded"]
                assert 0.003 < ref.mean() < 0.98
                r.set_counting(False)
                assert np.array_equal(r.is_occluded(O, D, tmax), ref)            # wide walk + leftover list
                r.set_counting(True)
                assert np.array_equal(r.is_occluded(O, D, tmax), ref)            # the reference's walk
                r.set_counting(False)
        r.clear()
        r.render(host_api.RT_MODE_PATH, 0, 5)
        frames[wide] = r.accumulator().copy()
        r.clear()
        r.render(host_api.RT_MODE_WHITTED, 0, 1)
        frames[wide + "w"] = r.accumulator().copy()
        r.close()
    for wide in ("1", "8"):
        assert np.array_equal(frames["0"].view(np.uint32), frames[wide].view(np.uint32)), wide
        assert np.array_equal(frames["0w"].view(np.uint32), frames[wide + "w"].view(np.uint32)), wide


@pytest.mark.parametrize("wide8", ["0", "1"])
def test_wide_walk_after_refit(wide8, scenes, oracle_api, host_api, monkeypatch):
    """rt_set_time deforms and refits the scene BVH; the wide nodes' boxes must follow (k_wide_sync); the quantised 8-wide
    nodes were rounded around the uploaded geometry and are dropped for the 4-wide ones."""
    monkeypatch.setenv("RT_WIDE", "1")
    monkeypatch.setenv("RT_WIDE8", wide8)
    o, orr, r, d = make_pair(scenes.mixed_small, oracle_api, host_api, 48, 32)
    O, D = random_rays(20000, 3)
    for t in (0.7, 3.1, 0.0):
        o.set_time(t)
        r.scene.set_time(t)
        for tmax in (None, np.full(len(O), 5.0, np.float32)):
            assert np.array_equal(r.is_occluded(O, D, tmax), o.is_occluded(O, D, tmax)["occluded"])
    r.close()


def test_config1_primary_hits(scenes, oracle_api, host_api):
    """BASELINE config 1: ico.obj scene, 256x256, primary rays only -> objIdx and t maps."""
    o, orr, r, d = make_pair(scenes.config1, oracle_api, host_api, 256, 256)
    obj_ref, t_ref, _ = orr.primary_hits(1e-6)
    obj, t = r.primary_hits(1e-6)
    assert np.array_equal(obj, obj_ref)
    assert np.array_equal(t.view(np.uint32), t_ref.view(np.uint32))
    assert len(np.unique(obj_ref)) >= 5  # sky, floor, light, spheres, icosahedra all visible
    r.close()


@pytest.mark.parametrize("name,kw,w,h", [("pretty_tlas", {"n_instances": 8}, 960, 540), ("bigb_instanced", {"n": 16, "mesh": "lowBigB"}, 640, 360),
                                         ("bigb_instanced", {"n": 16, "mesh": "BigB"}, 960, 540),  # BASELINE config 5's scene: BigB.obj (11,830 triangles) x 16
                                         ("tower", {}, 960, 540)])  # BASELINE config 4's scene: the 51,200-triangle stand-in
def test_primary_hits_at_scale(name, kw, w, h, scenes, oracle_api, host_api):
    """The bench scene (and the config-5 layout) at a quarter of the bench resolution: objIdx and t of every
    primary ray, timed kernels (TLAS children the ray cannot reach are skipped) against the oracle's walk."""
    o, orr, r, d = make_pair(scenes.REGISTRY[name], oracle_api, host_api, w, h, **kw)
    obj_ref, t_ref, _ = orr.primary_hits(0.001)
    obj, t = r.primary_hits(0.001)
    assert np.array_equal(obj, obj_ref)
    assert np.array_equal(t.view(np.uint32), t_ref.view(np.uint32))
    assert (obj_ref >= 100).mean() > 0.02  # instanced triangles are in view
    r.close()


def check_frames(orr, r, mode, frames, host_api, tol=RADIANCE_TOL):
    orr.scene.set_raytracer(mode == "whitted")
    orr.clear()
    orr.render(0, frames, nthreads=0)
    ref = orr.accumulator()
    r.clear()
    r.render(host_api.RT_MODE_WHITTED if mode == "whitted" else host_api.RT_MODE_PATH, 0, frames)
    got = r.accumulator()
    err, cls_ok = rel_err(got[..., :3], ref[..., :3])
    assert cls_ok, "non-finite pixels differ by class"
    assert err.max() <= tol, "max relative radiance error %g" % err.max()
    it = 1 if mode == "whitted" else frames
    # resolved 8-bit pixels: identical wherever the float accumulators are identical
    same = (got == ref).all(-1) | (~np.isfinite(ref).all(-1))
    px, px_ref = r.resolve(it), orr.resolve(it)
    assert np.array_equal(px[same], px_ref[same])
    return err.max(), float((got[..., :3] == ref[..., :3]).mean())


RENDER_SCENES = [
    ("background", {}, 96, 64),
    ("mixed_small", {}, 96, 64),
    ("scene3", {"force_diffuse": False}, 96, 54),
    ("scene3", {"force_diffuse": True, "split": 3}, 96, 54),
    ("tlas_test2", {}, 96, 64),
    ("pretty_tlas", {"n_instances": 4}, 96, 54),
    ("pretty_tlas", {"n_instances": 8}, 240, 135),  # the bench scene (BASELINE config 3) at 1/8 size
    ("bigb_instanced", {"n": 16, "mesh": "lowBigB"}, 128, 72),  # BASELINE config 5's layout with the small mesh
    ("bigb_instanced", {"n": 16, "mesh": "BigB"}, 192, 108),  # BASELINE config 5's scene itself (189,280 instanced triangles) at 1/20 size
    ("tower", {}, 192, 108),  # BASELINE config 4's scene (51,200-triangle stand-in for eifel.obj + synthetic sky) at 1/10 size
    ("terrain", {"n": 160}, 160, 90),  # the out-of-cache profile's scene (profiles/out_of_cache.py runs n = 2048: 8.4 M triangles) at 51,200 triangles
]


@pytest.mark.parametrize("name,kw,w,h", RENDER_SCENES)
def test_whitted_frame(name, kw, w, h, scenes, oracle_api, host_api):
    o, orr, r, d = make_pair(scenes.REGISTRY[name], oracle_api, host_api, w, h, **kw)
    check_frames(orr, r, "whitted", 1, host_api)
    r.close()


@pytest.mark.parametrize("name,kw,w,h", RENDER_SCENES)
@pytest.mark.parametrize("frames", [1, 4, 16])
@pytest.mark.parametrize("pipeline", ["slot", "stream"])
def test_path_frames(name, kw, w, h, frames, pipeline, scenes, oracle_api, host_api, monkeypatch):
    # "stream": the dense wavefront (csrc/rt_stream.h, the default); "slot": the slot wavefront of csrc/rt_kernels.h (what batches
    # above the slot budget and RT_COUNT_REFERENCE launches run)
    monkeypatch.setenv("RT_STREAM", "1" if pipeline == "stream" else "0")
    o, orr, r, d = make_pair(scenes.REGISTRY[name], oracle_api, host_api, w, h, **kw)
    check_frames(orr, r, "path", frames, host_api)
    r.close()


@pytest.mark.parametrize("name,kw", [("mixed_small", {}), ("pretty_tlas", {"n_instances": 4})])
@pytest.mark.parametrize("w,h,frames", [(37, 23, 1), (37, 23, 3), (7, 5, 1), (65, 1, 1), (63, 1, 2), (129, 3, 5)])
def test_path_frames_ragged_batches(name, kw, w, h, frames, scenes, oracle_api, host_api, monkeypatch):
    """Batches whose sample count is not a multiple of 64 (and batches of less than one wave): the dense pipeline hands out
    positions per GROUP of 64 entries and its shading kernel counts inside the group with ballots (csrc/rt_stream.h k_assign /
    k_shade_s); the last group of such a batch is ragged, and the grid's last iteration reaches past the batch."""
    monkeypatch.setenv("RT_STREAM", "1")
    o, orr, r, d = make_pair(scenes.REGISTRY[name], oracle_api, host_api, w, h, **kw)
    check_frames(orr, r, "path", frames, host_api)
    r.close()


@pytest.mark.parametrize("name,kw", [("pretty_scene1", {}), ("pretty_animation", {}), ("bigb_scene", {}), ("christ_scene", {}), ("tlas_test", {}),
                                     ("scene1", {}), ("scene2", {}), ("scene4", {}), ("scene5", {}), ("scene6", {}), ("scene7", {"nx": 64, "ny": 64}),
                                     ("scene7", {})])
def test_reference_factories(name, kw, scenes, oracle_api, host_api):
    """Every remaining scene factory of the reference (template/scene.h:791-1209, as data in scenes.py): primary hits and
    their work tallies, one Whitted frame, two path frames.  scene7 at its full 255 x 255 spheres (65,025 materials):
    primary hits and the Whitted frame."""
    full7 = name == "scene7" and not kw
    o, orr, r, d = make_pair(scenes.REGISTRY[name], oracle_api, host_api, 96, 54, **kw)
    obj_ref, t_ref, cnt_ref = orr.primary_hits(1e-6)
    r.set_counting(True)
    r.counters()
    obj, t = r.primary_hits(1e-6)
    cnt = r.counters()
    r.set_counting(False)
    assert np.array_equal(obj, obj_ref)
    assert np.array_equal(t.view(np.uint32), t_ref.view(np.uint32))
    for k in ("inner_visits", "prim_tests", "tlas_inner", "instance_visits"):
        assert cnt[k] == cnt_ref[k], k
    check_frames(orr, r, "whitted", 1, host_api)
    if not full7:
        check_frames(orr, r, "path", 2, host_api)
    r.close()


def test_fisheye_camera(scenes, oracle_api, host_api):
    """Camera::GetPrimaryRay fisheye branch (camera.h:26-32): primary hits and a Whitted frame."""
    o, orr, r, d = make_pair(scenes.mixed_small, oracle_api, host_api, 64, 40)
    cam = orr.camera()
    for rr in (orr, r):
        rr.set_camera(cam[0], cam[1], cam[2], cam[3], fisheye=True, view_angle=0.4, y_angle=0.3)
    obj_ref, t_ref, _ = orr.primary_hits(1e-6)
    obj, t = r.primary_hits(1e-6)
    assert np.array_equal(obj, obj_ref)
    assert np.array_equal(t.view(np.uint32), t_ref.view(np.uint32))
    check_frames(orr, r, "whitted", 1, host_api)
    check_frames(orr, r, "path", 2, host_api)
    r.close()


def test_frames_one_by_one_equal_batch(scenes, oracle_api, host_api):
    """Progressive accumulation: 6 calls of one frame == one call of 6 frames (path regeneration)."""
    o, orr, r, d = make_pair(scenes.mixed_small, oracle_api, host_api, 64, 40)
    r.clear()
    for f in range(6):
        r.render(host_api.RT_MODE_PATH, f, 1)
    a = r.accumulator()
    r.clear()
    r.render(host_api.RT_MODE_PATH, 0, 6)
    b = r.accumulator()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    r.close()


def test_tiles_equal_full_frame(scenes, oracle_api, host_api):
    """Pixel-tile sharding: rendering row bands separately gives the full frame bit for bit
    (per-pixel RNG streams), which is what makes the multi-GPU shard exact."""
    o, orr, r, d = make_pair(scenes.mixed_small, oracle_api, host_api, 64, 48)
    for mode, frames in ((host_api.RT_MODE_WHITTED, 1), (host_api.RT_MODE_PATH, 3)):
        r.clear()
        r.render(mode, 0, frames)
        full = r.accumulator()
        r.clear()
        for y0, y1 in ((0, 7), (7, 30), (30, 48)):
            r.render(mode, 0, frames, y0=y0, y1=y1)
        assert np.array_equal(full.view(np.uint32), r.accumulator().view(np.uint32))
    r.close()


def test_trace_batch_and_renderer_surface(scenes, oracle_api, host_api):
    """Renderer::Trace / Sample on caller rays (rt_trace_batch), Renderer::Tick, and the single-ray
    Scene::FindNearest / IsOccluded forms."""
    o, orr, r, d = make_pair(scenes.mixed_small, oracle_api, host_api, 48, 32)
    O, D = orr.primary_rays()
    # Whitted on primary rays == one Whitted frame (no RNG involved)
    orr.scene.set_raytracer(True)
    orr.clear(); orr.render(0, 1, nthreads=0)
    ref = orr.accumulator()[..., :3].reshape(-1, 3)
    got = r.trace_batch(host_api.RT_MODE_WHITTED, O, D, depth=4)
    err, cls_ok = rel_err(got, ref)
    assert cls_ok and err.max() <= RADIANCE_TOL
    assert np.array_equal(r.trace_batch(host_api.RT_MODE_WHITTED, O[:5], D[:5], depth=0), np.zeros((5, 3), np.float32))
    one = r.trace_one(O[100], D[100], 4, path=False)
    assert np.array_equal(one, got[100])
    # the other combination of function and flag (renderer.cpp:33-43, 107-121, 143-153: unreachable from Tick, reachable through the
    # preserved surface) is answered by the general kernels, never by the other branch: test_trace_and_sample_with_the_other_flag
    assert np.isfinite(r.trace_one(O[100], D[100], 4, path=True)).all()
    # Tick in Whitted mode fills accumulator and pixels
    r.tick()
    acc = r.tick_accumulator()
    e2, c2 = rel_err(acc[..., :3].reshape(-1, 3), ref)
    assert c2 and e2.max() <= RADIANCE_TOL
    assert np.array_equal(r.tick_pixels(), r.resolve(1))
    # Tick in path mode: three ticks == three frames
    r.scene.set_raytracer(False)
    r.clear()
    for _ in range(3):
        r.tick()
    orr.scene.set_raytracer(False)
    orr.clear(); orr.render(0, 3, nthreads=0)
    e3, c3 = rel_err(r.tick_accumulator()[..., :3], orr.accumulator()[..., :3])
    assert c3 and e3.max() <= RADIANCE_TOL
    # single-ray forms
    for i in (0, 500, 1000, 1535):
        ref1 = o.find_nearest(O[i:i + 1], D[i:i + 1], t_min=1e-6)
        t, obj, n = r.scene.find_nearest_one(O[i], D[i], t_min=1e-6)
        assert obj == ref1["obj"][0] and np.float32(t) == ref1["t"][0]
        occ = o.is_occluded(O[i:i + 1], D[i:i + 1], np.array([3.0], np.float32))["occluded"][0]
        assert r.scene.is_occluded_one(O[i], D[i], 3.0) == bool(occ)
    r.close()


def test_tick_with_camera_change(scenes, oracle_api, host_api):
    """Renderer::Tick across a camera change in path mode (renderer.cpp:247-255, :273-275, :287-294): the frame on
    which the camera changed clears the accumulator, is resolved with the iteration count read BEFORE the reset and
    does not advance the count -- the displayed pixels and the count must follow the reference's bookkeeping."""
    o, orr, r, d = make_pair(scenes.mixed_small, oracle_api, host_api, 48, 32)
    r.scene.set_raytracer(False)
    o.set_raytracer(False)
    cam = orr.camera()
    moved = cam.copy()
    moved[0] += np.float32(0.25)  # camPos
    frame = 0
    for step, change in enumerate([False, False, False, True, False, False, True, True, False]):
        if change:
            c = moved if (step % 2) else cam
            r.set_camera(c[0], c[1], c[2], c[3])
            orr.set_camera(c[0], c[1], c[2], c[3])
        px_ref, it_ref = orr.tick(change, frame)
        r.tick()
        frame += 1
        assert r.iteration() == it_ref, step
        got, ref = r.tick_accumulator(), orr.accumulator()
        err, cls_ok = rel_err(got[..., :3], ref[..., :3])
        assert cls_ok and err.max() <= RADIANCE_TOL, step
        # the displayed pixels: identical where the accumulators are bit for bit (the gamma is evaluated in single precision since
        # round 4, so that is no longer most pixels), and never more than one step of an 8-bit channel apart anywhere else
        same = (got == ref).all(-1)
        assert same.mean() > 0.2
        px = r.tick_pixels()
        assert np.array_equal(px[same], px_ref[same]), step
        ch = lambda p: np.stack([(p >> 16) & 255, (p >> 8) & 255, p & 255], -1).astype(np.int32)
        assert np.abs(ch(px) - ch(px_ref)).max() <= 1, step
    r.close()


@pytest.mark.parametrize("name,kw,w,h,frames", [("mixed_small", {}, 96, 64, 4), ("pretty_tlas", {"n_instances": 8}, 240, 135, 3), ("scene3", {"force_diffuse": False}, 96, 54, 4),
                                               ("tower", {}, 120, 68, 2), ("bigb_instanced", {"n": 16, "mesh": "lowBigB"}, 128, 72, 2)])
def test_exact_gamma_switch(name, kw, w, h, frames, scenes, oracle_api, host_api, monkeypatch):
    """The product evaluates the gamma of a finished path sample in single precision (csrc/rt_kernels.h gamma_powf: <= 4 ulp from the
    reference's pow-in-double-rounded-to-float, an OUTPUT transform no ray or draw depends on).  RT_EXACT_GAMMA=1 restores the
    reference's expression: 'rounds like the reference' stays checkable.  With it, in both pipelines, the share of pixels
    whose accumulator equals the oracle's bit for bit is back where rounds 1-3 had it (above one half; the rest is the radiance's
    own last-ulp differences, which the gamma passes on), never below the fast form's, and the two forms agree to a few ulp."""
    out = {}
    for key, env in (("exact_stream", {"RT_EXACT_GAMMA": "1", "RT_STREAM": "1"}), ("exact_slot", {"RT_EXACT_GAMMA": "1", "RT_STREAM": "0"}), ("fast", {})):
        for k in ("RT_EXACT_GAMMA", "RT_STREAM"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        o, orr, r, d = make_pair(scenes.REGISTRY[name], oracle_api, host_api, w, h, **kw)
        assert ("exact_gamma=%d" % (1 if "RT_EXACT_GAMMA" in env else 0)) in r.build_info()
        err, same = check_frames(orr, r, "path", frames, host_api)
        out[key] = (err, same, r.accumulator().copy())
        r.close()
    print({k: (v[0], v[1]) for k, v in out.items()})
    assert np.array_equal(out["exact_stream"][2].view(np.uint32), out["exact_slot"][2].view(np.uint32))
    assert out["exact_stream"][1] > 0.5 and out["exact_stream"][1] >= out["fast"][1]
    e, _ = rel_err(out["fast"][2][..., :3], out["exact_stream"][2][..., :3])
    assert e.max() <= 2e-6  # the fast form: a few ulp per sample


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0], [0, 1], "all"])
def test_multi_context_renderer(devices, scenes, oracle_api, host_api):
    """rapt::Renderer over several contexts (one host thread + one rt_ctx per GPU, rows interleaved,
    device-to-device gather into context 0): Whitted and path Ticks must give the one-context frame bit for bit.
    [0, 0] / [0, 0, 0] run the whole multi-context path on one GPU; [0, 1] and "all" need more than one."""
    ndev = host_api.rt_lib().rt_device_count()
    if devices == "all":
        devices = list(range(ndev))
    if max(devices) >= ndev or (len(set(devices)) > 1 and ndev < 2) or len(devices) < 2:
        pytest.skip("needs %d HIP devices, %d visible" % (max(devices) + 1, ndev))
    w, h = 96, 61  # an odd height: the shares differ in size
    frames = {}
    for key, devs in (("one", None), ("many", devices)):
        r = host_api.HostRenderer(w, h, devices=devs)
        d = scenes.pretty_tlas(r.scene, n_instances=4)
        r.commit()
        c = d["camera"]
        r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
        r.tick()
        out = [r.tick_accumulator().copy(), r.tick_pixels().copy()]
        r.scene.set_raytracer(False)
        for _ in range(3):
            r.tick()
        out += [r.tick_accumulator().copy(), r.tick_pixels().copy(), r.iteration()]
        frames[key] = out
        r.close()
    a, b = frames["one"], frames["many"]
    assert np.isfinite(a[2][..., :3]).mean() > 0.5 and (a[2][..., :3] > 0).mean() > 0.5
    for k in range(4):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    assert a[4] == b[4]
    # Scene::SetTime must reach every context's copy of the geometry (each GPU refits its own tree): an animated
    # scene-BVH scene, rows interleaved over the contexts, equals the one-context frame at every time
    anim = {}
    for key, devs in (("one", None), ("many", devices)):
        r = host_api.HostRenderer(w, h, devices=devs)
        scenes.mixed_small(r.scene)
        r.commit()
        out = []
        for t in (0.0, 1.7, 4.2):
            r.scene.set_time(t)
            r.tick()
            out.append(r.tick_accumulator().copy())
        anim[key] = out
        r.close()
    # Q-learning through Tick (rapt::Renderer::EnableQLearning): the contexts add their integer reward sums after every path frame
    # and all apply the total, so the sharded Ticks equal the one-context Ticks and every context ends with the same table
    ql = {}
    for key, devs in (("one", None), ("many", devices)):
        r = host_api.HostRenderer(w, h, devices=devs)
        scenes.mixed_small(r.scene)
        r.commit()
        r.scene.set_raytracer(False)
        r.tick_qlearning(6, (-4, -1, -4), (4, 5, 6))
        for _ in range(4):
            r.tick()
        ql[key] = (r.tick_accumulator().copy(), r.qlearn_table().copy())
        r.close()
    assert np.array_equal(ql["one"][0].view(np.uint32), ql["many"][0].view(np.uint32))
    assert np.array_equal(ql["one"][1].view(np.uint32), ql["many"][1].view(np.uint32)) and ql["one"][1].min() < 0.9
    assert not np.array_equal(anim["one"][0], anim["one"][1])  # the geometry really moved
    for k in range(3):
        assert np.array_equal(anim["one"][k].view(np.uint32), anim["many"][k].view(np.uint32)), ("set_time", k)


def test_bench_two_ranks_equal_one(host_api):
    """bench.py --gpus 2 through torch.distributed.run (one process per rank, rows interleaved, accumulator gather to
    rank 0) must report the frame checksum of the one-rank run.  With two GPUs visible the ranks use the RCCL
    ("nccl") backend on a GPU each, exactly like the driver's scaling runs; on a one-GPU box the gloo rehearsal
    mode lets both ranks share the device."""
    import json, os, subprocess, sys
    from conftest import ROOT
    ndev = host_api.rt_lib().rt_device_count()
    common = ["--steps", "2", "--warmup", "1", "--width", "320", "--height", "181", "--spp", "4", "--no-cpu-baseline"]  # (a warm-up step: the first step of a process pays torch's and HIP's first-call costs, which are neither render nor gather)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    if ndev < 2:
        env["RAPT_DIST_BACKEND"] = "gloo"
    def last_json(out):
        return json.loads([l for l in out.decode().splitlines() if l.startswith("{")][-1])
    one = last_json(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, env=env, cwd=ROOT, timeout=600))
    two = last_json(subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                             "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, env=env, cwd=ROOT, timeout=600))
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["ranks_seen"] == 1 and two["ranks_seen"] == 2  # an all-reduce of 1 over the process group
    # the driver's own command shape: `python bench.py --gpus 2` with no WORLD_SIZE starts its ranks itself (a child process)
    env_plain = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    self_started = last_json(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, env=env_plain, cwd=ROOT, timeout=600))
    assert self_started["n_gpus"] == 2 and self_started["ranks_seen"] == 2 and self_started["frame_checksum"] == one["frame_checksum"]
    assert one["metric"] == two["metric"] == "Mrays/s at 320×181×4spp"
    assert one["frame_checksum"] == two["frame_checksum"]
    assert two["rays_per_step"] == one["rays_per_step"]  # the two shards trace exactly the rays of the whole frame
    # the render / gather split of a multi-rank step (rank 0 and the maximum over the ranks); null on one rank
    assert one["render_ms"] is None and one["gather_ms"] is None
    for key in ("render_ms", "gather_ms"):
        assert set(two[key]) == {"rank0", "max"} and 0 <= two[key]["rank0"] <= two[key]["max"], (key, two[key])
    assert two["render_ms"]["max"] + two["gather_ms"]["max"] >= 0.3 * two["ms_per_step"]  # the two parts are most of the step (the rest: clearing the accumulator, the fences)
    # one share of an N-rank run by itself (profiling lines: profiles/r04_shares_all_ranks.txt): any rank's rows
    share = last_json(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--emulate-world", "2", "--emulate-rank", "1"] + common, env=env, cwd=ROOT, timeout=600))
    assert "rank 1's rows of a 2-rank shard" in share["metric"]


def test_bench_line_witnesses_parity(host_api):
    """bench.py's cpu_baseline leg renders the step's frames with the oracle ONCE, times that as the baseline, and compares its
    accumulator with the device's as a whole: the JSON line carries parity_check, and the run fails above 1e-4."""
    import json, os, subprocess, sys
    from conftest import ROOT
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--width", "160", "--height", "90", "--spp", "3", "--cpu-seconds", "30"], cwd=ROOT, timeout=600)
    d = json.loads([l for l in out.decode().splitlines() if l.startswith("{")][-1])
    pc, cb = d["parity_check"], d["cpu_baseline"]
    assert pc["ok"] and pc["frames"] == 3 and pc["pixels"] == 160 * 90 and pc["max_rel_err"] <= 1e-4 and pc["nonfinite_class_equal"]
    assert pc["same_frames_as_timed_step"] and pc["same_checksum_as_timed_step"]
    assert cb["kind"] == "port" and 1 <= cb["cores"] <= cb["physical_cores"] and "frames 0..2" in cb["sample"]


def test_bench_line_carries_its_live_legs(host_api):
    """The legs bench.py runs outside the timed region of its default line, here on a small frame and a small terrain:
    the out-of-cache record (a scene built by rt_build_bvh_split on the device, SURVEY 8(d) bytes over the HIP-event kernel time, the crop's hits
    against the oracle's bit for bit), the eight 1/8 row shares with their row-set checksums and the projection, Renderer::Tick in both modes,
    the split of the run into legs, every rank's device."""
    import json, os, subprocess, sys
    from conftest import ROOT
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--width", "160", "--height", "96", "--spp", "3", "--cpu-seconds", "30",
                                   "--force-legs", "--ooc-n", "96", "--ooc-spp", "2", "--ooc-steps", "2"], cwd=ROOT, timeout=600)
    d = json.loads([l for l in out.decode().splitlines() if l.startswith("{")][-1])
    r = d["roofline"]
    assert r["bound"].startswith("vector-memory") and r["unit"] == "TA busy fraction" and r["pmc"]["used"] is False and r["frac"] is None  # another size: the counter file is not quoted
    o = r["hbm"]["out_of_cache_live"]
    assert "rt_build_bvh_split on the device" in o["scene"] and o["steps"] == 2 and o["kernel_ms"] > 0 and o["algorithmic_bytes_per_launch"] > 0 and o["algorithmic_GBps"] > 0
    assert o["crop_parity"]["bit_exact"] and o["crop_parity"]["rays"] == 4096 and o["crop_parity"]["hits"] > 0
    sh = d["share_ms"]
    assert sh["world"] == 8 and len(sh["per_rank"]) == 8 and len(set(sh["row_set_checksums"])) == 8 and sh["slowest"] == max(sh["per_rank"]) and sh["projected_speedup"] > 0
    tk = d["tick_ms"]
    assert tk["whitted"] > 0 and tk["path"] > 0 and tk["whitted_pixels_checksum"] != tk["path_pixels_checksum"]
    assert d["gpu_leg_s"] > 0 and d["cpu_leg_s"] > 0 and d["timed_region_s"] > 0 and len(d["ranks_devices"]) == 1 and ":" in d["ranks_devices"][0]
    assert d["parity_check"]["ok"] and d["cpu_baseline"]["kind"] == "port"


@pytest.mark.parametrize("name,kw", [("tlas_test2", {"mesh": "BigB"}), ("pretty_tlas", {"n_instances": 4}), ("mixed_small", {})])
def test_members_below_scene_level(name, kw, scenes, oracle_api, host_api):
    """The surface SURVEY.md 8b lists under Scene: bvh::Intersect / IsOccluded (bvh.h:57, :65), tlas::Intersect /
    IsOccluded (tlas.h:20-21), bvhInstance::BIntersect / IsOccluded (bvhInstance.h:11-12) and Scene::GetSkyColor
    (template/scene.h:1312) -- batch forms through rt_intersect_scope / rt_occluded_scope / rt_sky_color_batch and the
    one-ray members of the C++ mirror -- and Trace / Sample with an energy other than float3(1)."""
    o, orr, r, d = make_pair(scenes.REGISTRY[name], oracle_api, host_api, 48, 32, **kw)
    O, D = random_rays(12000, 21, center=(0.0, 1.0, 3.0), spread=5.0)
    pO, pD = orr.primary_rays()
    O, D = np.concatenate([O, pO]), np.concatenate([D, pD])
    tmax = np.random.default_rng(2).uniform(0.2, 30.0, len(O)).astype(np.float32)
    scopes = [(1, 0), (2, 0)] + ([(3, 0), (3, o.n_instances - 1), (2, r.scene.blas_count() - 1)] if d["tlas"] else [])
    for scope, index in scopes:
        for tm in (None, tmax):
            ref = o.scope_nearest(scope, index, O, D, tm)
            got = r.scope_nearest(scope, index, O, D, tm)
            assert np.array_equal(got["obj"], ref["obj"]), (scope, index)
            hit = ref["obj"] != -1
            assert np.array_equal(got["t"][hit].view(np.uint32), ref["t"][hit].view(np.uint32))
            assert np.array_equal(got["normal"][hit].view(np.uint32), ref["normal"][hit].view(np.uint32))
            assert np.array_equal(got["mat"][hit], ref["mat"][hit])
            assert np.array_equal(r.scope_occluded(scope, index, O, D, tm), o.scope_occluded(scope, index, O, D, tm))
        assert (o.scope_nearest(scope, index, O, D)["obj"] != -1).mean() > 0.002
    # the one-ray members of the host mirror
    for i in (0, 777, 4000, len(O) - 50):
        for which, index, scope in ([(0, 0, 2), (1, 0, 1), (2, o.n_instances - 1, 3)] if d["tlas"] else [(0, 0, 2)]):
            ref1 = o.scope_nearest(scope, index, O[i:i + 1], D[i:i + 1])
            t, obj, n = r.scene.member_intersect(which, index, O[i], D[i])
            assert obj == ref1["obj"][0]
            if obj != -1:
                assert np.float32(t) == ref1["t"][0] and np.array_equal(n, ref1["normal"][0])
            assert r.scene.member_occluded(which, index, O[i], D[i], 9.0) == bool(o.scope_occluded(scope, index, O[i:i + 1], D[i:i + 1], np.array([9.0], np.float32))[0])
    # Scene::GetSkyColor
    sky_ref = o.sky_color(D)
    assert np.array_equal(r.sky_color(D).view(np.uint32), sky_ref.view(np.uint32))
    assert np.array_equal(r.scene.sky_color_one(D[5]), sky_ref[5])
    # Trace / Sample with a caller energy
    e = (0.5, 0.75, 0.25)
    for mode in (host_api.RT_MODE_WHITTED, host_api.RT_MODE_PATH):
        o.set_raytracer(mode == host_api.RT_MODE_WHITTED)
        ref = orr.trace_rays(mode, pO, pD, 4, e, seed_base=99)
        got = r.trace_batch(mode, pO, pD, depth=4, seed_base=99, energy=e)
        err, cls_ok = rel_err(got, ref)
        assert cls_ok and err.max() <= RADIANCE_TOL
        if mode == host_api.RT_MODE_WHITTED:
            r.scene.set_raytracer(True)
            e1, c1 = rel_err(r.trace_one(pO[200], pD[200], 4, path=False, energy=e), orr.trace_rays(0, pO[200:201], pD[200:201], 4, e, seed_base=0x12345678)[0])
            assert c1 and e1.max() <= RADIANCE_TOL
    r.close()


@pytest.mark.parametrize("name,kw", [("mixed_small", {}), ("scene3", {"force_diffuse": False}), ("tlas_test2", {}), ("shiny", {"shininess": 0.25})])
def test_trace_and_sample_with_the_other_flag(name, kw, scenes, oracle_api, host_api):
    """Renderer::Trace with scene.raytracer == false (Russian roulette at every surface hit, sampled light positions,
    diffuse::scatter's hemisphere draw and the indirect child it feeds: renderer.cpp:33-43, 107-121) and Renderer::Sample with the flag
    set (roulette, unsampled lights: :143-153).  Tick never makes these calls; the preserved Trace / Sample surface can.  One lane per
    call tree on the device (k_trace_general, k_sample_general), held against the oracle's restatement of the same lines on the camera
    rays of a frame: radiance within the tolerance, non-finite values equal by class.  Materials are built with raytracer == false, as a
    scene constructed while the flag is clear has them (template/scene.h:605-620 draws only then)."""
    w, h = 48, 32
    # ("shiny": a floor with shinieness != 0 -- the mirror child inside the light loop, renderer.cpp:100-103 / :172-173, on top of everything else)
    build = (lambda b: _shiny_scene(b, kw["shininess"], False)) if name == "shiny" else (lambda b: scenes.REGISTRY[name](b, rt=False, **kw))
    o = oracle_api.OracleScene()
    d = build(o)
    r = host_api.HostRenderer(w, h)
    build(r.scene)
    r.commit()
    orr = oracle_api.OracleRenderer(o, w, h)
    O, D = orr.primary_rays()
    e = (0.9, 0.8, 0.7)
    for mode, flag in ((host_api.RT_MODE_WHITTED, False), (host_api.RT_MODE_PATH, True)):
        o.set_raytracer(flag)
        r.scene.set_raytracer(flag)
        for depth in (1, 2, 4):
            ref = orr.trace_rays(mode, O, D, depth, e, seed_base=4242)
            r.set_scene_raytracer(1 if flag else 0)
            got = r.trace_batch(mode, O, D, depth=depth, seed_base=4242, energy=e)
            err, cls_ok = rel_err(got, ref)
            assert cls_ok and err.max() <= RADIANCE_TOL, (name, mode, depth, err.max())
            assert np.abs(ref[np.isfinite(ref)]).sum() > 0
        # the C++ surface sets the flag from its Scene: one ray through rapt::Renderer::Trace / Sample
        one = r.trace_one(O[700], D[700], 4, path=(mode == host_api.RT_MODE_PATH), energy=e)
        ref1 = orr.trace_rays(mode, O[700:701], D[700:701], 4, e, seed_base=0x12345678)[0]
        e1, c1 = rel_err(one, ref1)
        assert c1 and e1.max() <= RADIANCE_TOL
    # and the flag can be handed back to the function: Sample as Tick calls it
    r.set_scene_raytracer(-1)
    o.set_raytracer(False)
    ref = orr.trace_rays(host_api.RT_MODE_PATH, O, D, 4, e, seed_base=9)
    got = r.trace_batch(host_api.RT_MODE_PATH, O, D, depth=4, seed_base=9, energy=e)
    err, cls_ok = rel_err(got, ref)
    assert cls_ok and err.max() <= RADIANCE_TOL
    r.close()


def test_edge_cases(scenes, oracle_api, host_api):
    r = host_api.HostRenderer(16, 8)
    # calls before a scene is uploaded fail loudly
    with pytest.raises(RuntimeError):
        r.render(host_api.RT_MODE_WHITTED)
    # planes only (no triangles, no spheres): root is a single leaf
    s = r.scene
    m = s.diffuse(0.8, (1, 1, 1), 0.0, 1.0, 4)
    s.plane(0, m, (0, 1, 0), 0)
    s.area_light(11, (0, 5, 0), 10.0, (1, 1, 1), 1.0, (0, -1, 0))
    s.build(0)
    r.commit()
    o = oracle_api.OracleScene()
    mo = o.diffuse(0.8, (1, 1, 1), 0.0, 1.0, 4)
    o.plane(0, mo, (0, 1, 0), 0)
    o.area_light(11, (0, 5, 0), 10.0, (1, 1, 1), 1.0, (0, -1, 0))
    o.build(0)
    O, D = random_rays(512, 5)
    ref = o.find_nearest(O, D)
    got = r.find_nearest(O, D)
    assert np.array_equal(got["obj"], ref["obj"]) and np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))
    # empty batches are accepted
    assert len(r.find_nearest(np.zeros((0, 3)), np.zeros((0, 3)))["t"]) == 0
    assert len(r.is_occluded(np.zeros((0, 3)), np.zeros((0, 3)))) == 0
    # a shiny diffuse material: Whitted takes a mirror branch per visible light (pending-branch stack);
    # path mode switches to the one-lane-per-sample kernel (see test_general_path_kernel)
    r2 = host_api.HostRenderer(16, 8)
    s2 = r2.scene
    m2 = s2.diffuse(0.8, (1, 1, 1), 0.6, 0.4, 4, shininess=0.5)
    s2.plane(0, m2, (0, 1, 0), 0)
    s2.area_light(11, (0, 5, 0), 10.0, (1, 1, 1), 1.0, (0, -1, 0))
    s2.build(0)
    r2.commit()
    r2.render(host_api.RT_MODE_WHITTED)  # Whitted handles it (mirror branch per visible light)
    o2 = oracle_api.OracleScene()
    mo2 = o2.diffuse(0.8, (1, 1, 1), 0.6, 0.4, 4, shininess=0.5)
    o2.plane(0, mo2, (0, 1, 0), 0)
    o2.area_light(11, (0, 5, 0), 10.0, (1, 1, 1), 1.0, (0, -1, 0))
    o2.build(0)
    orr2 = oracle_api.OracleRenderer(o2, 16, 8)
    orr2.render(0, 1)
    err, cls_ok = rel_err(r2.accumulator()[..., :3], orr2.accumulator()[..., :3])
    assert cls_ok and err.max() <= RADIANCE_TOL
    r.close(); r2.close()


@pytest.mark.parametrize("config", ["config2", "config3", "config4", "config5"])
def test_full_size_properties(config, scenes, oracle_api, host_api):
    """Every 1-GPU / multi-GPU BASELINE configuration at its full resolution (1920x1080; config 5: 3840x2160), where the
    oracle is too slow to run whole frames: run-to-run determinism, two-way shard == whole frame bit for bit
    (contiguous bands and the interleaved rows the multi-GPU path uses), and oracle parity on five sampled rows."""
    probe = oracle_api.OracleScene()
    cfg = scenes.REGISTRY[config](probe)
    probe.close()
    w, h = cfg["width"], cfg["height"]
    o, orr, r, d = make_pair(scenes.REGISTRY[config], oracle_api, host_api, w, h)
    r.clear(); r.render(host_api.RT_MODE_PATH, 0, 2)
    a = r.accumulator()
    r.clear(); r.render(host_api.RT_MODE_PATH, 0, 2)
    assert np.array_equal(a.view(np.uint32), r.accumulator().view(np.uint32))  # run-to-run deterministic
    r.clear()
    r.render(host_api.RT_MODE_PATH, 0, 2, y0=0, y1=h // 2)
    r.render(host_api.RT_MODE_PATH, 0, 2, y0=h // 2, y1=h)
    assert np.array_equal(a.view(np.uint32), r.accumulator().view(np.uint32))  # two-way shard == whole
    r.clear()
    r.render_rows(host_api.RT_MODE_PATH, 0, 2, 0, 2, (h + 1) // 2)
    r.render_rows(host_api.RT_MODE_PATH, 0, 2, 1, 2, h // 2)
    assert np.array_equal(a.view(np.uint32), r.accumulator().view(np.uint32))  # interleaved rows (rank r of 2) == whole
    rows = [0, h // 3 - 27, h // 2, (3 * h) // 4 - 33, h - 1]
    orr.scene.set_raytracer(False)
    for y in rows:
        orr.render(0, 2, y0=y, y1=y + 1, nthreads=0)
    ref = orr.accumulator()
    err, cls_ok = rel_err(a[rows][..., :3], ref[rows][..., :3])
    assert cls_ok and err.max() <= RADIANCE_TOL
    assert (ref[rows][..., :3] > 0).mean() > 0.5  # the rows show something
    r.close()


def test_full_size_config5_with_the_sampler_on(scenes, oracle_api, host_api):
    """BASELINE config 5 as stated -- BigB.obj x 16 through bvhInstance / TLAS at 3840x2160, "Q-learning sampler on" (grid 16^3,
    every fourth sample pays rewards: what bench.py --workload config5 --qlearn runs) -- at its full resolution, two batches of
    two frames with a table update between them.  PARITY UNPINNED like the sampler itself.  Held: run-to-run
    determinism of the second batch's frame, reward sums and the learned table; the interleaved two-way row shard (rank r of 2)
    gives the same frame AND the same integer sums in both batches; and the oracle, loaded with the table the device learned
    in batch 1 (rt_qlearn_get_table -> orc_qlearn_set_table), renders five rows of batch 2 within the radiance tolerance."""
    probe = oracle_api.OracleScene()
    cfg = scenes.REGISTRY["config5"](probe)
    probe.close()
    w, h = cfg["width"], cfg["height"]
    box = ((-12.0, -2.0, -8.0), (12.0, 10.0, 16.0))
    o, orr, r, d = make_pair(scenes.REGISTRY["config5"], oracle_api, host_api, w, h)
    orr.scene.set_raytracer(False)

    def two_batches(sharded):
        r.qlearn_enable(16, box[0], box[1], 0.3, 0.2, 1.0, 3)  # a fresh table
        out = []
        for b in range(2):
            r.clear()
            if sharded:
                r.render_rows(host_api.RT_MODE_PATH, 2 * b, 2, 0, 2, (h + 1) // 2)
                r.render_rows(host_api.RT_MODE_PATH, 2 * b, 2, 1, 2, h // 2)
            else:
                r.render(host_api.RT_MODE_PATH, 2 * b, 2)
            sums, cnts = r.qlearn_sums()
            r.qlearn_apply()
            out.append((r.accumulator().copy(), sums, cnts, r.qlearn_table()))
        return out
    first, again, shards = two_batches(False), two_batches(False), two_batches(True)
    for b in range(2):
        assert first[b][2].sum() > 0  # rewards were paid
        for other in (again, shards):
            assert np.array_equal(first[b][0].view(np.uint32), other[b][0].view(np.uint32)), b  # the frame
            assert np.array_equal(first[b][1], other[b][1]) and np.array_equal(first[b][2], other[b][2]), b  # integer sums and counts
            assert np.array_equal(first[b][3].view(np.uint32), other[b][3].view(np.uint32)), b  # the table after the update
    table1 = first[0][3]
    assert table1.min() < 0.9 and table1.max() > 1.1  # batch 1 taught it something
    orr.qlearn_enable(16, box[0], box[1], 0.3, 0.2, 1.0, 3)
    orr.qlearn_set_table(table1)
    rows = [0, h // 3 - 27, h // 2, (3 * h) // 4 - 33, h - 1]
    for y in rows:
        orr.render(2, 2, y0=y, y1=y + 1, nthreads=0)
    ref, a = orr.accumulator(), first[1][0]
    err, cls_ok = rel_err(a[rows][..., :3], ref[rows][..., :3])
    assert cls_ok and err.max() <= RADIANCE_TOL, err.max()
    assert (ref[rows][..., :3] > 0).mean() > 0.5
    r.close()


def test_gamma_of_finished_samples(host_api):
    """The gamma of a finished path-mode sample (renderer.cpp:279-282: pow(c, 1 / 2.2) per channel, a double-precision pow rounded
    to float in the reference) is evaluated in SINGLE precision since round 4 (csrc/rt_kernels.h gamma_powf): an
    output transform that feeds no ray, no texel index and no random draw.  Its error against the rounded double-precision value
    is held here on all 256 values a sky texel can take -- a camera ray that leaves the scene is finished with sky / 255 -- at
    <= 4 ulp, and the two places that apply it (the 256-entry table of k_generate_s, k_accumulate for samples stored raw) must
    give the same bits."""
    g32 = np.float32(0.57142857142857142857143)
    frames = {}
    for lut in ("1", "0"):
        os.environ["RT_GAMMA_LUT"] = lut
        try:
            r = host_api.HostRenderer(8, 8)
            s = r.scene
            m = s.diffuse(0.8, (1, 1, 1))
            s.sphere(0, m, (0, -10, 0), 1.0)  # below the camera, which looks straight up: every camera ray leaves the scene
            s.area_light(11, (0, -20, 0), 1.0, (1, 1, 1), 0.5, (0, -1, 0))
            got = []
            for b0 in range(0, 256, 3):
                tex = np.zeros((2, 4, 3), np.uint8)
                tex[...] = [b0, min(b0 + 1, 255), min(b0 + 2, 255)]
                s.sky(tex)
                s.build(0)
                r.commit()
                r.set_camera((0, 0, 0), (-0.5, 2, 0.5), (0.5, 2, 0.5), (-0.5, 2, -0.5))
                r.clear()
                r.render(host_api.RT_MODE_PATH, 0, 1)
                a = r.accumulator()[..., :3]
                assert np.array_equal(a.view(np.uint32), np.broadcast_to(a[0, 0], a.shape).view(np.uint32))  # every pixel is the sky's colour
                got.append(a[0, 0].copy())
            frames[lut] = np.concatenate(got)[:256]
            r.close()
        finally:
            del os.environ["RT_GAMMA_LUT"]
    assert np.array_equal(frames["0"].view(np.uint32), frames["1"].view(np.uint32))  # table and k_accumulate: one function
    c = (np.arange(256, dtype=np.float32) / np.float32(255)).astype(np.float32)
    ref = np.power(c.astype(np.float64), np.float64(g32)).astype(np.float32)
    ulp = np.spacing(np.maximum(ref, np.float32(1e-30)))
    err = np.abs(frames["1"].astype(np.float64) - ref.astype(np.float64)) / ulp
    assert err.max() <= 4.0, (err.max(), int(err.argmax()))


def test_limits_are_reported(scenes, oracle_api, host_api):
    """Device-path limits surface as errors, never as silent fallbacks: > 32 lights, and rendering
    rows outside the image."""
    r = host_api.HostRenderer(16, 8)
    s = r.scene
    m = s.diffuse(0.8, (1, 1, 1))
    s.plane(0, m, (0, 1, 0), 0)
    for i in range(33):
        s.area_light(11 + i, (i, 5, 0), 10.0, (1, 1, 1), 1.0, (0, -1, 0))
    s.build(0)
    with pytest.raises(RuntimeError, match="lights"):
        r.commit()
    r.close()
    r = host_api.HostRenderer(16, 8)
    scenes.mixed_small(r.scene)
    r.commit()
    with pytest.raises(RuntimeError, match="rows"):
        r.render(host_api.RT_MODE_PATH, 0, 1, y0=4, y1=12)
    with pytest.raises(RuntimeError):
        r.render(host_api.RT_MODE_WHITTED, 0, 2)  # Whitted frames overwrite: one at a time
    r.close()


def _many_lights(b, n=12):
    """a floor, a diffuse and a glass sphere, a metal mesh, n area lights on a ring (more than the 8 the device path stopped at before round 3)"""
    import math
    fl = b.diffuse(0.8, (1, 1, 1), 0.0, 1.0, 4)
    df = b.diffuse(0.8, (0.2, 0.9, 0.3), 0.6, 0.4, 10)
    gl = b.glass(1.5, (0.8, 0.9, 1.0))
    me = b.metal(0.7, (1.0, 0.8, 0.3))
    for i in range(n):
        a = 2 * math.pi * i / n
        b.area_light(11 + i, (2.5 * math.cos(a), 3.0 + 0.1 * i, 1.0 + 2.5 * math.sin(a)), 3.0, (1.0, 0.9 - 0.03 * i, 0.7 + 0.02 * i), 0.6, (0, -1, 0))
    b.sphere(1, df, (0.4, 0.5, 1.2), 0.5)
    b.sphere(2, gl, (-0.8, 0.4, 0.6), 0.4)
    b.mesh_obj(3, assets_mod.obj_path("ico"), me, (1.3, 0.6, 0.4), 0.5)
    b.plane(0, fl, (0, 1, 0), 0)
    b.build(0)
    return dict(name="many_lights", tlas=False)


def test_more_than_eight_lights(scenes, oracle_api, host_api):
    """The reference holds its lights in a vector without a bound (template/scene.h:1374); the device path sizes the per-light planes
    of its path state by the scene's count.  Twelve lights: primary hits, Whitted and path frames against the oracle."""
    global assets_mod
    import importlib
    assets_mod = importlib.import_module("ray-and-pathtracer_amd.assets")
    o, orr, r, d = make_pair(_many_lights, oracle_api, host_api, 80, 48)
    obj_ref, t_ref, _ = orr.primary_hits(1e-6)
    obj, t = r.primary_hits(1e-6)
    assert np.array_equal(obj, obj_ref) and np.array_equal(t.view(np.uint32), t_ref.view(np.uint32))
    check_frames(orr, r, "whitted", 1, host_api)
    check_frames(orr, r, "path", 6, host_api)
    r.close()


def test_deep_tree_uses_the_spill_stack(oracle_api, host_api):
    """A degenerate LONGESTAXIS tree over geometrically spaced triangles is deeper than the 15 stack
    entries kept in LDS: the global spill part of the traversal stack must give the same hits."""
    n = 40
    tris = []
    for i in range(n):
        x = 2.0 ** i * 1e-6
        tris.append([x, 0, 1, x, 1, 1, x * 1.1, 0, 1])
    tris = np.array(tris, dtype=np.float32)
    def fill(s):
        m = s.diffuse(0.8, (1, 1, 1))
        s.mesh_raw(1, m, tris)
        s.build(2)  # LONGESTAXIS: halves the extent, peeling off one triangle per level
    o = oracle_api.OracleScene(); fill(o)
    r = host_api.HostRenderer(8, 8); fill(r.scene); r.commit()
    assert o.bvh_dump(-1)["max_depth"] > 20
    rng = np.random.default_rng(0)
    O = np.stack([rng.uniform(0, 600, 4000), rng.uniform(0, 1, 4000), np.zeros(4000)], 1).astype(np.float32)
    O[:, 0] = np.exp(rng.uniform(np.log(1e-6), np.log(600.0), 4000)).astype(np.float32)
    D = np.tile(np.array([[0, 0, 1]], np.float32), (4000, 1))
    D[:, 0] = rng.uniform(-0.05, 0.05, 4000).astype(np.float32)
    D /= np.linalg.norm(D, axis=1, keepdims=True).astype(np.float32)
    ref = o.find_nearest(O, D)
    got = r.find_nearest(O, D)
    assert (ref["obj"] != -1).sum() > 20
    assert np.array_equal(got["obj"], ref["obj"]) and np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))
    assert np.array_equal(r.is_occluded(O, D), o.is_occluded(O, D)["occluded"])
    r.close()


@pytest.mark.parametrize("name,kw", [("pretty_tlas", {"n_instances": 8}), ("pretty_tlas", {"n_instances": 3}), ("tlas_test2", {})])
def test_tlas_copy_in_lds_equals_global_walk(name, kw, scenes, oracle_api, host_api, monkeypatch):
    """A TLAS that leaves at least 8 stack rows of a traversal block's LDS (28 words per pair + 13 per instance; ~50
    instances) is walked in every block's own LDS copy (csrc/rt_scene_dev.h trace_persistent, RT_TLAS_LDS read at
    rt_upload_scene); larger ones and RT_TLAS_LDS=0 keep the records in global memory.  Same hits, same occlusion
    answers, same frames, and the oracle's."""
    out = {}
    for lds in ("1", "0"):
        monkeypatch.setenv("RT_TLAS_LDS", lds)
        o, orr, r, d = make_pair(scenes.REGISTRY[name], oracle_api, host_api, 96, 54, **kw)
        if lds == "1":
            O, D = random_rays(20000, 5, center=(0.5, 1.0, 2.0), spread=6.0)
            ref = o.find_nearest(O, D)
            occ = o.is_occluded(O, D)["occluded"]
        got = r.find_nearest(O, D)
        assert np.array_equal(got["obj"], ref["obj"]) and np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))
        assert np.array_equal(r.is_occluded(O, D), occ)
        frames = {}
        for mode, n in ((host_api.RT_MODE_WHITTED, 1), (host_api.RT_MODE_PATH, 3)):
            r.clear()
            r.render(mode, 0, n)
            frames[mode] = r.accumulator().copy()
        out[lds] = frames
        r.close()
    for mode in out["1"]:
        assert np.array_equal(out["1"][mode].view(np.uint32), out["0"][mode].view(np.uint32))


def test_zero_hash_stream(scenes, oracle_api, host_api):
    """The stream index whose Wang hash is 0 (xorshift32's fixed point): same replacement state on
    both sides, no endless rejection loop in the hemisphere sampler."""
    o, orr, r, d = make_pair(scenes.mixed_small, oracle_api, host_api, 16, 8)
    base = 1768515948 - 40
    orr.scene.set_raytracer(False)
    orr.render(0, 2, seed_base=base, nthreads=0)
    r.render(host_api.RT_MODE_PATH, 0, 2, seed_base=base)
    err, cls_ok = rel_err(r.accumulator()[..., :3], orr.accumulator()[..., :3])
    assert cls_ok and err.max() <= RADIANCE_TOL
    r.close()


def test_cpp_host_example(tmp_path, scenes, oracle_api, host_api):
    """examples/render_scene.cpp: a C++ host using rapt::Renderer / Scene (LoadFile, Commit, Tick) over
    the C ABI produces the pixels the oracle's Tick loop resolves to."""
    import os, subprocess
    from conftest import ROOT, pkg
    sf = pkg("scene_file")
    exe = str(tmp_path / "render_scene")
    host, csrc = os.path.join(ROOT, "ray-and-pathtracer_amd", "host"), os.path.join(ROOT, "ray-and-pathtracer_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "examples", "render_scene.cpp"), "-I" + host, "-L" + host, "-lrapt_host",
                           "-L" + csrc, "-lrt_amd", "-Wl,-rpath," + host, "-Wl,-rpath," + csrc, "-o", exe])
    scene_path = str(tmp_path / "mixed.rapt")
    w = sf.SceneWriter(scene_path)
    scenes.mixed_small(w)
    o = oracle_api.OracleScene()
    scenes.mixed_small(o)
    for mode, frames, devs in (("whitted", 1, None), ("path", 3, None), ("path", 3, "0,0")):
        out = str(tmp_path / (mode + ".ppm"))
        subprocess.check_call([exe, scene_path, out, "48", "32", mode, str(frames)] + ([devs] if devs else []))  # "0,0": two contexts on device 0
        raw = open(out, "rb").read()
        px = np.frombuffer(raw[raw.index(b"255\n") + 4:], dtype=np.uint8).reshape(32, 48, 3)
        o.set_raytracer(mode == "whitted")
        orr = oracle_api.OracleRenderer(o, 48, 32)
        orr.render(0, frames, nthreads=0)
        ref = orr.resolve(1 if mode == "whitted" else frames)
        ref_rgb = np.stack([(ref >> 16) & 255, (ref >> 8) & 255, ref & 255], -1).astype(np.int32)
        # 8-bit values may differ by one where the float accumulators differ in the last bits
        assert np.abs(px.astype(np.int32) - ref_rgb).max() <= 1
        assert (px.astype(np.int32) == ref_rgb).mean() > 0.99
        orr.close()


@pytest.mark.parametrize("name,kw", [("mixed_small", {}), ("mixed_small", {"split": 3}), ("scene3", {"force_diffuse": False}), ("tower", {})])
def test_animation_and_refit(name, kw, scenes, oracle_api, host_api):
    """Scene::SetTime with animation on (template/scene.h:1228-1244) + bvh::Refit (bvh.cpp:556-594),
    both on the GPU (rt_set_time): hits, distances, normals and traversal counters still equal the
    oracle's for the deformed, refitted scene; t = 0 restores the uploaded geometry."""
    o, orr, r, d = make_pair(scenes.REGISTRY[name], oracle_api, host_api, 48, 32, **kw)
    O, D = random_rays(4000, 11)
    pO, pD = orr.primary_rays()
    O, D = np.concatenate([O, pO]), np.concatenate([D, pD])
    base = o.find_nearest(O, D)
    for t in (0.6, 2.9, 7.5, 0.0):
        o.set_time(t)
        r.scene.set_time(t)
        ref = o.find_nearest(O, D)
        r.set_counting(True); r.counters()
        got = r.find_nearest(O, D)
        cnt = r.counters(); r.set_counting(False)
        assert np.array_equal(got["obj"], ref["obj"])
        assert np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))
        hit = ref["obj"] != -1
        assert np.array_equal(got["normal"][hit].view(np.uint32), ref["normal"][hit].view(np.uint32))
        for k in ("inner_visits", "prim_tests"):
            assert cnt[k] == ref["counters"][k], k
        if t != 0.0:
            assert not np.array_equal(ref["t"], base["t"])  # the geometry really moved
        else:
            assert np.array_equal(ref["t"].view(np.uint32), base["t"].view(np.uint32))
        tmax = np.full(len(O), 6.0, np.float32)
        assert np.array_equal(r.is_occluded(O, D, tmax), o.is_occluded(O, D, tmax)["occluded"])
    o.set_time(1.3); r.scene.set_time(1.3)
    check_frames(orr, r, "whitted", 1, host_api)
    r.close()


def test_set_time_refuses_tlas(scenes, oracle_api, host_api):
    r = host_api.HostRenderer(16, 8)
    scenes.tlas_test2(r.scene)
    r.commit()
    with pytest.raises(RuntimeError, match="TLAS"):
        r.scene.set_time(1.0)
    r.close()


def _shiny_scene(b, shininess, rt):
    """mixed_small with a shiny floor / materials built with raytracer == false: the cases where
    Renderer::Sample's random draws interleave with occlusion queries."""
    from conftest import pkg
    assets = pkg("assets")
    b.sky(assets.synthetic_sky(64, 32, seed=4))
    b.area_light(11, (1.0, 4.0, 1.0), 10.0, (1, 1, 1), 1.0, (0, -1, 0))
    b.area_light(12, (-1.0, 3.0, 0.5), 5.0, (1, 1, 1), 0.5, (0, -1, 0))
    gl = b.glass(1.5, (0.6, 0.6, 1.0), (0.1, 0.2, 0.05), rt=rt)
    me = b.metal(0.7, (0.85, 0.65, 0.12), rt=rt)
    df = b.diffuse(0.8, (0, 1, 0), 0.6, 0.4, 10, rt=rt)
    fl = b.diffuse(0.8, (1, 1, 1), 0.3, 0.7, 4, shininess=shininess, rt=rt)
    b.mesh_obj(1, assets.obj_path("ico"), gl, (-0.9, 0.6, 0.6), 0.5)
    b.mesh_obj(2, assets.obj_path("stellatedDode"), me, (0.9, 0.7, 0.8), 0.5)
    b.mesh_obj(3, assets.obj_path("three"), df, (0.0, 0.5, 1.8), 1.2)
    b.sphere(1, gl, (0.2, 0.35, 0.2), 0.35)
    b.plane(0, fl, (0, 1, 0), 0)
    b.build(0)
    return dict(name="shiny", tlas=False)


@pytest.mark.parametrize("shininess,rt", [(0.4, True), (0.0, False), (0.25, False)])
def test_general_path_kernel(shininess, rt, scenes, oracle_api, host_api):
    """Path mode with a shiny diffuse floor and/or materials built with raytracer == false: the
    one-lane-per-sample kernel (k_sample_general) must reproduce Renderer::Sample's depth-first draw order."""
    fn = lambda b: _shiny_scene(b, shininess, rt)
    o, orr, r, d = make_pair(fn, oracle_api, host_api, 72, 48)
    check_frames(orr, r, "path", 3, host_api)
    check_frames(orr, r, "whitted", 1, host_api)
    # Renderer::Sample on caller rays through the same kernel
    O, D = orr.primary_rays()
    got = r.trace_batch(host_api.RT_MODE_PATH, O[:512], D[:512], depth=4, seed_base=77)
    assert np.isfinite(got).mean() > 0.8
    r.close()


def _silhouette_rays(o, n_inst, rng, per_inst=400):
    """Rays aimed at the vertices that DEFINE each instance's world box (and at random vertices), from
    near and far origins, with unit and non-unit directions: the cases where a conservative box test
    could wrongly drop an instance."""
    Os, Ds = [], []
    for i in range(n_inst):
        I = o.instance_dump(i)
        T = I["T"].reshape(4, 4).astype(np.float64)
        tris, _ = o.mesh_tris(I["blas"])
        ok = np.isfinite(tris[:, 9:12]).all(1)
        v = tris[ok, :9].reshape(-1, 3).astype(np.float64)
        w = v @ T[:3, :3].T + T[:3, 3]
        pick = np.concatenate([w.argmin(0), w.argmax(0), rng.integers(0, len(w), per_inst - 6)])
        tgt = w[pick]
        dist = rng.choice([0.5, 3.0, 30.0, 1e3, 1e5], len(tgt))
        dirs = rng.normal(size=(len(tgt), 3))
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        org = (tgt - dirs * dist[:, None]).astype(np.float32)
        d = tgt - org.astype(np.float64)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        d *= rng.choice([1.0, 1.0, 1e-3, 37.0], len(tgt))[:, None]
        # nudge a third of the directions by a few ulps so that they graze the vertex on either side
        d = d.astype(np.float32)
        nud = rng.integers(-3, 4, d.shape).astype(np.int32)
        nud[::3] = 0
        d = (d.view(np.int32) + nud).view(np.float32)
        Os.append(org), Ds.append(d)
    return np.concatenate(Os), np.concatenate(Ds)


@pytest.mark.parametrize("name,kw", [("tlas_test2", {}), ("pretty_tlas", {"n_instances": 8}), ("bigb_instanced", {"n": 9, "mesh": "lowBigB"})])
def test_tlas_reach_culling_keeps_results(name, kw, scenes, oracle_api, host_api):
    """The timed kernels (counting off) drop TLAS children whose hittable geometry the ray cannot reach
    (csrc/rt_scene_dev.h, reach[]); hits, t, normals and occlusion must stay bit-identical to the
    reference walk, which enters every instance whose (much larger) bounds the ray touches."""
    o, orr, r, d = make_pair(scenes.REGISTRY[name], oracle_api, host_api, 64, 48, **kw)
    rng = np.random.default_rng(77)
    n_inst = o.n_instances
    sO, sD = _silhouette_rays(o, n_inst, rng)
    rO, rD = random_rays(20000, 5, center=(0.0, 1.0, 5.0), spread=8.0)
    pO, pD = orr.primary_rays()
    O, D = np.concatenate([sO, rO, pO]), np.concatenate([sD, rD, pD])
    r.set_counting(False)
    for t_min in (1e-6, 0.001):
        ref = o.find_nearest(O, D, t_min=t_min)
        got = r.find_nearest(O, D, t_min=t_min)
        assert np.array_equal(got["obj"], ref["obj"])
        assert np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))
        hit = ref["obj"] != -1
        assert np.array_equal(got["mat"][hit & (ref["mat"] >= 0)], ref["mat"][hit & (ref["mat"] >= 0)])
        assert np.array_equal(got["normal"][hit].view(np.uint32), ref["normal"][hit].view(np.uint32))
    # the silhouette rays must really hit instanced geometry often, or the test proves nothing
    ref_s = o.find_nearest(sO, sD)
    assert (ref_s["obj"] != -1).mean() > 0.5
    for tmax in (None, np.full(len(O), 4.0, dtype=np.float32), rng.uniform(0.5, 2e5, len(O)).astype(np.float32)):
        assert np.array_equal(r.is_occluded(O, D, tmax), o.is_occluded(O, D, tmax)["occluded"])
    # counting on walks without culling: same answers, and the reference's tallies
    r.set_counting(True)
    r.counters()
    got = r.find_nearest(O, D)
    cnt = r.counters()
    r.set_counting(False)
    ref = o.find_nearest(O, D)
    assert np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))
    for k in ("inner_visits", "prim_tests", "tlas_inner", "instance_visits"):
        assert cnt[k] == ref["counters"][k], k
    # RT_COUNT_EXECUTED: the tallies of the culled walk -- same answers, never more work than the reference's
    r.set_counting(host_api.RT_COUNT_EXECUTED)
    r.counters()
    got = r.find_nearest(O, D)
    cnt2 = r.counters()
    r.set_counting(False)
    assert np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))
    assert cnt2["rays_nearest"] == cnt["rays_nearest"] and cnt2["prim_tests"] <= cnt["prim_tests"]
    assert cnt2["instance_visits"] < cnt["instance_visits"] and cnt2["inner_visits"] < cnt["inner_visits"] and cnt2["tlas_inner"] <= cnt["tlas_inner"]


def _fuzz_rays(o, n_inst, rng, n, center, extent):
    """A chunk of n rays, a quarter each of: random rays through the scene's volume; rays aimed at instance-box
    silhouette vertices from 0.5 .. 1e5 units away with +-3 ulp nudges; rays from far origins (up to the reach
    tables' origin limit and beyond) towards the scene; near-degenerate directions (one or two components tiny or
    exactly zero, non-unit lengths)."""
    q = n // 4
    O = np.empty((n, 3), np.float32)
    D = np.empty((n, 3), np.float32)
    # random
    O[:q] = (rng.uniform(-1, 1, (q, 3)) * extent + center).astype(np.float32)
    D[:q] = rng.normal(size=(q, 3)).astype(np.float32)
    # silhouette
    sO, sD = _silhouette_rays(o, n_inst, rng, per_inst=max(8, q // n_inst))
    k = min(q, len(sO))
    O[q:q + k], D[q:q + k] = sO[:k], sD[:k]
    if k < q:
        O[q + k:2 * q] = (rng.uniform(-1, 1, (q - k, 3)) * extent + center).astype(np.float32)
        D[q + k:2 * q] = rng.normal(size=(q - k, 3)).astype(np.float32)
    # far origins aimed at the scene
    dist = np.exp(rng.uniform(np.log(10.0), np.log(1e6), q))[:, None]
    dirs = rng.normal(size=(q, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    tgt = rng.uniform(-1, 1, (q, 3)) * extent * 0.6 + center
    O[2 * q:3 * q] = (tgt - dirs * dist).astype(np.float32)
    D[2 * q:3 * q] = (tgt - O[2 * q:3 * q].astype(np.float64)).astype(np.float32)
    # near-degenerate directions
    O[3 * q:] = (rng.uniform(-1, 1, (n - 3 * q, 3)) * extent + center).astype(np.float32)
    dd = rng.normal(size=(n - 3 * q, 3))
    kill = rng.integers(0, 3, len(dd))
    dd[np.arange(len(dd)), kill] *= rng.choice([0.0, 1e-30, 1e-12, 1e-7, 1e-4], len(dd))
    two = rng.uniform(size=len(dd)) < 0.3
    dd[two, (kill[two] + 1) % 3] *= rng.choice([0.0, 1e-20, 1e-6], int(two.sum()))
    dd *= rng.choice([1.0, 1.0, 1e-3, 250.0], len(dd))[:, None]
    D[3 * q:] = dd.astype(np.float32)
    return O, D


@pytest.mark.parametrize("name,kw,total", [("pretty_tlas", {"n_instances": 8}, 100_000_000), ("bigb_instanced", {"n": 16, "mesh": "BigB"}, 12_000_000),
                                           ("tlas_test2", {"mesh": "BigB"}, 12_000_000)])
def test_reach_cull_fuzz_gpu_vs_gpu(name, kw, total, scenes, oracle_api, host_api, monkeypatch):
    """1e8 rays on the bench scene (1.2e7 on two more): the timed traversal (TLAS children whose reach box the ray
    misses are dropped, DESIGN.md section 4 finding 8) against the same kernels walking like the reference
    (RT_COUNT_REFERENCE: no culling) -- GPU against GPU, because this one equivalence needs volume, not an oracle.
    Hit ids, t, materials and normals of Scene::FindNearest and the flags of Scene::IsOccluded must be bit-identical."""
    if name == "bigb_instanced":
        monkeypatch.setenv("RT_WIDE", "1")  # the two smaller runs also put the wide occlusion walks against the binary one: 4-wide exact boxes ...
    if name == "tlas_test2":
        monkeypatch.setenv("RT_WIDE8", "1")  # ... and 8-wide quantised boxes
    o, orr, r, d = make_pair(scenes.REGISTRY[name], oracle_api, host_api, 16, 8, **kw)
    rng = np.random.default_rng(20260)
    n_inst = o.n_instances
    boxes = np.stack([o.instance_dump(i)["bounds"] for i in range(n_inst)])
    lo, hi = boxes[:, :3].min(0), boxes[:, 3:].max(0)
    hi = np.minimum(hi, 50.0)  # a .tri mesh's sentinel triangles stretch the reference's boxes to 999 (Q10)
    center, extent = (lo + hi) / 2, (hi - lo) / 2 + 1.0
    chunk = 4_000_000
    done, hits, diff_near, diff_occ = 0, 0, 0, 0
    while done < total:
        n = min(chunk, total - done)
        O, D = _fuzz_rays(o, n_inst, rng, n, center, extent)
        tmax = np.where(rng.uniform(size=n) < 0.5, np.float32(1e34), rng.uniform(0.1, 40.0, n)).astype(np.float32)
        r.set_counting(False)
        a = r.find_nearest(O, D, tmax, t_min=0.001)
        oa_ = r.is_occluded(O, D, tmax)
        r.set_counting(host_api.RT_COUNT_REFERENCE)
        b = r.find_nearest(O, D, tmax, t_min=0.001)
        ob = r.is_occluded(O, D, tmax)
        r.set_counting(False)
        same = (a["obj"] == b["obj"]) & (a["t"].view(np.uint32) == b["t"].view(np.uint32)) & (a["mat"] == b["mat"]) & \
               (a["normal"].view(np.uint32) == b["normal"].view(np.uint32)).all(1)
        diff_near += int((~same).sum())
        diff_occ += int((oa_ != ob).sum())
        hits += int((b["obj"] >= 100).sum())
        done += n
    r.counters()
    assert diff_near == 0 and diff_occ == 0, (diff_near, diff_occ)
    assert hits > total // 50  # the rays do reach instanced triangles
    r.close()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1023, 1025, 4097, 70001])
def test_batch_sizes_cover_the_work_heads(n, scenes, oracle_api, host_api):
    """The traversal kernels cut a queue into 16 sub-queues of a multiple of 64 entries and hand them out
    in pieces of 64..256 (csrc/rt_scene_dev.h): every entry must be traced exactly once for any length."""
    o, orr, r, d = make_pair(scenes.REGISTRY["tlas_test2"], oracle_api, host_api, 32, 24)
    O, D = random_rays(n, 1000 + n, center=(0.0, 1.0, 3.0), spread=5.0)
    ref = o.find_nearest(O, D)
    r.set_counting(True)
    r.counters()
    got = r.find_nearest(O, D)
    cnt = r.counters()
    r.set_counting(False)
    assert cnt["rays_nearest"] == n
    assert np.array_equal(got["obj"], ref["obj"]) and np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))
    for k in ("inner_visits", "prim_tests", "tlas_inner", "instance_visits"):
        assert cnt[k] == ref["counters"][k], k
    assert np.array_equal(r.is_occluded(O, D), o.is_occluded(O, D)["occluded"])


@pytest.mark.parametrize("slots", ["777", "4096", "20000"])
@pytest.mark.parametrize("name,kw", [("mixed_small", {}), ("pretty_tlas", {"n_instances": 4})])
def test_fewer_slots_than_samples(slots, name, kw, scenes, oracle_api, host_api, monkeypatch):
    """With fewer slots than samples a slot whose sample is finished pulls the next one from the pool
    (k_finish); by default every sample of a batch has its own slot, so this path needs RT_SLOTS.  The
    frames must not depend on the slot count: same accumulator bits as with one slot per sample."""
    o, orr, r, d = make_pair(scenes.REGISTRY[name], oracle_api, host_api, 96, 54, **kw)
    r.clear()
    r.render(host_api.RT_MODE_PATH, 0, 12)
    full = r.accumulator()
    monkeypatch.setenv("RT_SLOTS", slots)  # (read when a context is created: csrc/rt_ctx.h Knobs)
    o2, orr2, r2, d2 = make_pair(scenes.REGISTRY[name], oracle_api, host_api, 96, 54, **kw)
    monkeypatch.delenv("RT_SLOTS")
    err, _ = check_frames(orr2, r2, "path", 12, host_api)
    assert np.array_equal(r2.accumulator().view(np.uint32), full.view(np.uint32))
    # Whitted: pending branches + pool
    r2.clear()
    r2.render(host_api.RT_MODE_WHITTED, 0, 1)
    few = r2.accumulator()
    r.clear()
    r.render(host_api.RT_MODE_WHITTED, 0, 1)
    assert np.array_equal(r.accumulator().view(np.uint32), few.view(np.uint32))
    r.close(), r2.close()


@pytest.mark.parametrize("name,kw,w,h,frames", [("pretty_tlas", {"n_instances": 4}, 160, 90, 7), ("mixed_small", {}, 96, 64, 5), ("bigb_instanced", {"n": 9, "mesh": "lowBigB"}, 128, 72, 3),
                                               ("tower", {}, 120, 68, 3), ("scene3", {}, 100, 60, 4), ("background", {}, 90, 60, 3)])
def test_stream_pipeline_equals_slot_pipeline(name, kw, w, h, frames, scenes, oracle_api, host_api, monkeypatch):
    """The dense path pipeline (csrc/rt_stream.h, the default: entries of a round are the survivors of the round before,
    every producer writes to compacted positions; RT_FUSE=0 runs it one kernel at a time) and its producer-side ray
    decisions (RT_DECIDE: generate / shade answer a ray whose first traversal step leaves nothing to visit), against the slot pipeline (RT_STREAM=0): identical accumulator bits -- per sample the arithmetic is the same, only where the path
    state lives differs -- the oracle's frame, identical frames from row shards, identical Sample() values for
    caller-supplied rays at every depth, and the same number of FindNearest / IsOccluded queries."""
    out, rays = {}, {}
    pO = pD = None
    for key, env in (("slot", {"RT_STREAM": "0"}), ("stream", {}), ("stream_serial", {"RT_FUSE": "0"}), ("stream_two_streams", {"RT_FUSE": "2"}), ("stream_one_launch_per_round", {"RT_FUSE": "1"}), ("stream_nodecide", {"RT_DECIDE": "0"}),
                     ("stream_gamma_at_the_store", {"RT_DEFER_GAMMA": "0"})):
        for k in ("RT_STREAM", "RT_FUSE", "RT_DECIDE", "RT_DEFER_GAMMA"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        o, orr, r, d = make_pair(scenes.REGISTRY[name], oracle_api, host_api, w, h, **kw)
        r.set_counting(host_api.RT_COUNT_EXECUTED); r.counters()
        r.clear()
        r.render(host_api.RT_MODE_PATH, 0, frames)
        near, occl = r.counters_split()
        r.set_counting(False)
        rays[key] = (near["rays_nearest"], occl["rays_occluded"], near["light_tests"], near["brute_tests"])
        counted = r.accumulator().copy()
        r.clear()
        r.render(host_api.RT_MODE_PATH, 0, frames)
        out[key] = r.accumulator().copy()
        assert np.array_equal(counted.view(np.uint32), out[key].view(np.uint32))  # counting launches render the same frame
        if key == "stream":
            check_frames(orr, r, "path", frames, host_api)
            r.clear()
            r.render_rows(host_api.RT_MODE_PATH, 0, frames, 0, 2, (h + 1) // 2)
            r.render_rows(host_api.RT_MODE_PATH, 0, frames, 1, 2, h // 2)
            assert np.array_equal(r.accumulator().view(np.uint32), out[key].view(np.uint32))
        if key in ("slot", "stream"):
            if pO is None:
                pO, pD = orr.primary_rays()
                pO, pD = pO[::7].copy(), pD[::7].copy()
            out[key + "_sample"] = [r.trace_batch(host_api.RT_MODE_PATH, pO, pD, depth, 99) for depth in (0, 1, 4)]
        r.close()
    for key in ("stream", "stream_serial", "stream_two_streams", "stream_one_launch_per_round", "stream_nodecide", "stream_gamma_at_the_store"):
        assert np.array_equal(out["slot"].view(np.uint32), out[key].view(np.uint32)), key
        assert rays[key] == rays["slot"], (key, rays[key], rays["slot"])
    for other in ("stream_sample",):
        for a, b in zip(out["slot_sample"], out[other]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), other


@pytest.mark.parametrize("name,kw,w,h", [("mixed_small", {}, 96, 64), ("pretty_tlas", {"n_instances": 4}, 160, 90), ("scene3", {"force_diffuse": False}, 96, 54),
                                         ("background", {}, 120, 80), ("tower", {}, 120, 68), ("bigb_instanced", {"n": 9, "mesh": "lowBigB"}, 128, 72)])
def test_whitted_single_launch_equals_rounds(name, kw, w, h, scenes, oracle_api, host_api, monkeypatch):
    """Renderer::Trace as ONE persistent launch per frame (csrc/rt_mega.h, the default: a lane keeps its pixel and runs the
    body of Trace at each hit inside the traversal kernel's flush) against the wavefront rounds (RT_MEGA=0, csrc/rt_kernels.h):
    the same accumulator bits (segments in the same depth-first order, terms added in the same order), the oracle's frame,
    row shards equal to the frame, and identical Trace() values for caller-supplied rays at every depth."""
    out = {}
    for mega in ("0", "1"):
        monkeypatch.setenv("RT_MEGA", mega)
        o, orr, r, d = make_pair(scenes.REGISTRY[name], oracle_api, host_api, w, h, **kw)
        if mega == "1":
            check_frames(orr, r, "whitted", 1, host_api)
        r.clear()
        r.render(host_api.RT_MODE_WHITTED, 0, 1)
        out[mega] = [r.accumulator().copy()]
        r.clear()
        r.render_rows(host_api.RT_MODE_WHITTED, 0, 1, 0, 2, (h + 1) // 2)
        r.render_rows(host_api.RT_MODE_WHITTED, 0, 1, 1, 2, h // 2)
        assert np.array_equal(r.accumulator().view(np.uint32), out[mega][0].view(np.uint32))
        pO, pD = orr.primary_rays()
        pO, pD = pO[::5].copy(), pD[::5].copy()
        for depth in (1, 2, 4, 6):
            out[mega].append(r.trace_batch(host_api.RT_MODE_WHITTED, pO, pD, depth, 7, energy=(0.9, 0.8, 0.7)))
        r.close()
    for a, b in zip(out["0"], out["1"]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("name,kw,w,h", [("mixed_small", {}, 200, 83), ("pretty_tlas", {"n_instances": 4}, 192, 108)])
def test_whitted_longest_first_order(name, kw, w, h, scenes, oracle_api, host_api, monkeypatch):
    """The single-launch Whitted frame (RT_MEGA_LEVELS=0) deals its pixels out by what they cost in the launch before
    (csrc/rt_mega.h "longest first": k_mega_hist / k_mega_order; frames of 16384 samples and more).  Any order must give the same
    frame: the first launch (no history: the multiplicative permutation), the second and third (history), row shards with their
    own histories, RT_MEGA_LPT=0, the wavefront rounds (RT_MEGA=0) and the default -- one launch per tree level with the terms
    logged under depth-first keys and added by k_whitted_reduce ("levels"; "auto", the default, times both forms on the first
    batches of a shape and keeps the faster) -- all leave the same accumulator bits.  200 x 83 is
    not a multiple of the 8-pixel tile nor of the padded queue."""
    frames = {}
    for key, env in (("rounds", {"RT_MEGA": "0"}), ("plain", {"RT_MEGA_LEVELS": "0", "RT_MEGA_LPT": "0"}), ("lpt", {"RT_MEGA_LEVELS": "0"}),
                     ("levels", {"RT_MEGA_LEVELS": "1"}), ("auto", {})):
        for k in ("RT_MEGA", "RT_MEGA_LPT", "RT_MEGA_LEVELS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = host_api.HostRenderer(w, h)
        d = scenes.REGISTRY[name](r.scene, **kw)
        r.commit()
        if "camera" in d:
            c = d["camera"]
            r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
        got = []
        for _ in range(3):
            r.clear()
            r.render(host_api.RT_MODE_WHITTED, 0, 1)
            got.append(r.accumulator().copy())
        for _ in range(2):  # shards: each call has another sample range, the history follows the last one
            r.clear()
            r.render_rows(host_api.RT_MODE_WHITTED, 0, 1, 0, 1, h // 2)
            r.render_rows(host_api.RT_MODE_WHITTED, 0, 1, h // 2, 1, h - h // 2)
            got.append(r.accumulator().copy())
        for _ in range(2):  # the same shard twice in a row: its history is used
            r.clear()
            r.render_rows(host_api.RT_MODE_WHITTED, 0, 1, 0, 1, h)
            got.append(r.accumulator().copy())
        r.close()
        for g in got[1:]:
            assert np.array_equal(g.view(np.uint32), got[0].view(np.uint32)), key
        frames[key] = got[0]
    for key in ("plain", "lpt", "levels", "auto"):
        assert np.array_equal(frames[key].view(np.uint32), frames["rounds"].view(np.uint32)), key


@pytest.mark.parametrize("name,kw,w,h", [("mixed_small", {}, 97, 61), ("pretty_tlas", {"n_instances": 4}, 120, 67), ("scene3", {"force_diffuse": False}, 96, 54),
                                         ("bigb_instanced", {"n": 9, "mesh": "lowBigB"}, 101, 57)])
def test_whitted_levels_depths_and_batches(name, kw, w, h, scenes, host_api, monkeypatch):
    """Renderer::Trace by tree levels (csrc/rt_mega.h k_whitted_level / k_whitted_reduce, RT_MEGA_LEVELS=1) against the
    wavefront rounds (RT_MEGA=0) at every depth from 1 (no children at all) to 7 (more levels than the default 4; keys of
    seven digits), for two frames (Whitted frames overwrite the accumulator: one per call), full frames and interleaved row
    shards: identical accumulator bits."""
    out = {}
    for key, env in (("rounds", {"RT_MEGA": "0"}), ("levels", {"RT_MEGA_LEVELS": "1"}), ("levels_overflow", {"RT_MEGA_LEVELS": "1", "RT_LEVEL_CAP": "256"})):
        for k in ("RT_MEGA", "RT_MEGA_LEVELS", "RT_LEVEL_CAP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = host_api.HostRenderer(w, h)
        d = scenes.REGISTRY[name](r.scene, **kw)
        r.commit()
        if "camera" in d:
            c = d["camera"]
            r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
        got = []
        for depth in (1, 2, 3, 4, 7):
            for frame0 in (5, 6):
                r.clear()
                r.render(host_api.RT_MODE_WHITTED, frame0, 1, max_depth=depth)
                got.append(r.accumulator().copy())
            r.clear()
            r.render_rows(host_api.RT_MODE_WHITTED, 5, 1, 0, 3, (h + 2) // 3, max_depth=depth)
            r.render_rows(host_api.RT_MODE_WHITTED, 5, 1, 1, 3, (h + 1) // 3, max_depth=depth)
            r.render_rows(host_api.RT_MODE_WHITTED, 5, 1, 2, 3, h // 3, max_depth=depth)
            got.append(r.accumulator().copy())
        r.close()
        out[key] = got
    for i, (a, b) in enumerate(zip(out["rounds"], out["levels"])):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), i
    # queues of 256 segments: every frame with more children than that overflows them and is repeated as one launch
    for i, (a, b) in enumerate(zip(out["rounds"], out["levels_overflow"])):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), i
    for k in range(0, len(out["levels"]), 3):  # the shards of a depth give its single frame
        assert np.array_equal(out["levels"][k].view(np.uint32), out["levels"][k + 2].view(np.uint32)), k


def test_qlearning_sampler_is_unbiased(scenes, oracle_api, host_api):
    """The guided sampler draws the indirect bounce with density 16 P / pi instead of the uniform hemisphere's 1 / (2 pi) and
    weighs it with 1 / (16 P) instead of 2: the same integral.  Held in LINEAR radiance -- rt_trace_batch returns raw Sample()
    values -- on a scene lit by the sky and by DirectionalLights only (scenes.qlearn_probe): an area light's disk can be hit, a
    path that does is +inf (Q7), and the guided sampler aims at the lights, so on the reference's own scenes the mean of the
    finite paths is not the same quantity for the two samplers (measured on the oracle: -2 % at 6 sigma on tlas_test2, entirely
    from the direct term of the paths dropped as infinite).  After five batches of learning, the mean luminance of Sample() over
    the frame's camera rays and 16 seeds agrees with the uniform-hemisphere sampler's within 4 standard errors of the
    difference, and that standard error is below 0.5 % of the mean: a 1 / (16 P) wrong by two per cent would fail.  The
    oracle's statement of the sampler must give the same numbers (same arithmetic: within the radiance tolerance)."""
    w, h, box = 64, 40, ((-4, -1, -4), (4, 5, 6))
    o, orr, r, d = make_pair(scenes.REGISTRY["qlearn_probe"], oracle_api, host_api, w, h)
    orr.scene.set_raytracer(False)
    r.qlearn_enable(8, box[0], box[1], 0.3, 0.2, 1.0, 0)
    orr.qlearn_enable(8, box[0], box[1], 0.3, 0.2, 1.0, 0)
    for b in range(5):
        r.render(host_api.RT_MODE_PATH, 3 * b, 3); r.qlearn_apply()
        orr.render(3 * b, 3, nthreads=0); orr.qlearn_apply()
    assert np.array_equal(orr.qlearn_state()[2].view(np.uint32), r.qlearn_table().view(np.uint32))
    tab = r.qlearn_table()
    assert tab.min() < 0.5 and tab.std() > 0.05  # it learned something (no reward exceeds the sky's 1 here, so nothing grows)
    pO, pD = orr.primary_rays()

    def seed_means(trace):
        out = []
        for k in range(16):
            v = trace(0x5EED0000 + 7919 * k).astype(np.float64)
            assert np.isfinite(v).all()
            out.append((0.2126 * v[:, 0] + 0.7152 * v[:, 1] + 0.0722 * v[:, 2]).mean())
        return np.array(out)
    g = seed_means(lambda sb: r.trace_batch(host_api.RT_MODE_PATH, pO, pD, 4, sb))
    g_ref = seed_means(lambda sb: orr.trace_rays(1, pO, pD, 4, (1, 1, 1), sb))
    r.qlearn_disable()
    u = seed_means(lambda sb: r.trace_batch(host_api.RT_MODE_PATH, pO, pD, 4, sb))
    se = np.sqrt(g.var(ddof=1) / len(g) + u.var(ddof=1) / len(u))
    assert se <= 0.005 * u.mean(), (se, u.mean())
    assert abs(g.mean() - u.mean()) <= 4 * se, (g.mean(), u.mean(), se)
    assert np.allclose(g, g_ref, rtol=RADIANCE_TOL, atol=0)
    r.close()


def test_qlearning_sampler_variance_falls_with_learning(scenes, oracle_api, host_api):
    """What guiding does to the estimator's variance, measured (nothing did): per-ray variance of Sample() over 24 seeds
    on the probe scene (sky + directional lights: every path is finite), linear radiance through rt_trace_batch.  The honest
    state of this sampler (profiles/r05_qlearn_variance.txt): with an untrained table it is ~5x the uniform-hemisphere sampler's
    (64 constant-density patches of the whole sphere, an eps / 64 floor under rarely drawn patches), learning brings it down
    (2.2x after 40 frames at 320x200) -- it does not get below the uniform sampler's on a scene lit by the whole sky.  Held here:
    equal means (test_qlearning_sampler_is_unbiased) and a variance that FALLS as the table learns."""
    w, h, box = 160, 100, ((-4, -1, -4), (4, 5, 6))
    o, orr, r, d = make_pair(scenes.REGISTRY["qlearn_probe"], oracle_api, host_api, w, h)
    pO, pD = orr.primary_rays()
    pO, pD = pO[::5].copy(), pD[::5].copy()

    def variance():
        def lum(sb):
            v = r.trace_batch(host_api.RT_MODE_PATH, pO, pD, 4, sb).astype(np.float64)
            assert np.isfinite(v).all()
            return 0.2126 * v[:, 0] + 0.7152 * v[:, 1] + 0.0722 * v[:, 2]
        return np.stack([lum(0x5EED0000 + 7919 * k) for k in range(24)]).var(axis=0, ddof=1).mean()
    vu = variance()
    r.qlearn_enable(8, box[0], box[1], 0.3, 0.2, 1.0, 0)
    r.render(host_api.RT_MODE_PATH, 0, 4); r.qlearn_apply()
    v1 = variance()
    for b in range(1, 12):
        r.render(host_api.RT_MODE_PATH, 4 * b, 4); r.qlearn_apply()
    v12 = variance()
    print("per-ray variance: uniform %.5f, guided after 1 batch %.5f (x%.2f), after 12 batches %.5f (x%.2f)" % (vu, v1, v1 / vu, v12, v12 / vu))
    assert v12 < 0.8 * v1 and v12 < 6 * vu, (vu, v1, v12)
    r.close()


def test_qlearning_reward_words_report_overflow(scenes, host_api):
    """A reward is ONE packed 64-bit atomic (count << 44 | 48.16 sum, csrc/rt_qlearn.h q_reward); the words are folded into the wide
    sums at the end of every batch of frames of a render call.  A count field past half its range must come back as RT_E_OVERFLOW
    from that call, never wrap silently: a 1-cell grid with every sample paying puts a whole 1080p batch on 64 words.  Two sizes:
    64 frames leave the busiest count inside [2^19, 2^20) (the fold sees it); 128 frames (~1.9 M rewards per word) carry it past
    2^20, where the field has wrapped by the time it is folded -- the lanes that paid into a word with bit 63 set saw it."""
    w, h = 1920, 1080
    r = host_api.HostRenderer(w, h)
    scenes.REGISTRY["mixed_small"](r.scene)
    r.commit()
    r.qlearn_enable(1, (-50, -50, -50), (50, 50, 50), 0.3, 0.2, 1.0, 0)
    r.clear(); r.render(host_api.RT_MODE_PATH, 0, 2)
    sums, cnts = r.qlearn_sums()  # two frames: well inside the range, and exact
    assert 0 < cnts.max() < (1 << 19) and cnts.sum() > w * h // 4
    per_frame = int(cnts.max()) // 2
    r.qlearn_apply()
    assert (1 << 19) <= per_frame * 64 < (1 << 20) < per_frame * 128, per_frame
    for frames in (64, 128):  # ~15 k rewards per word and frame
        with pytest.raises(RuntimeError, match="rewards for one"):
            r.render(host_api.RT_MODE_PATH, 2, frames)
        r.qlearn_enable(1, (-50, -50, -50), (50, 50, 50), 0.3, 0.2, 1.0, 0)  # a fresh table
    r.close()


def test_qlearning_sums_in_caller_memory(scenes, host_api):
    """rt_qlearn_bind_sums: the pending reward sums live in the caller's device arrays -- what one process per
    GPU all-reduces in place over RCCL (bench.py --qlearn with several ranks) instead of four host copies per exchange.  Held here on
    one GPU: the bound tensors ARE the sums (equal to an unbound renderer's rt_qlearn_get_sums after the same frames), doubling them
    in place is what two identical ranks' all-reduce leaves, and the table learnt from them equals rt_qlearn_set_sums' of the same."""
    import torch
    w, h, box = 64, 40, ((-4, -1, -4), (4, 5, 6))
    rs = []
    for _ in range(2):
        r = host_api.HostRenderer(w, h)
        scenes.REGISTRY["mixed_small"](r.scene)
        r.commit()
        r.qlearn_enable(8, box[0], box[1], 0.3, 0.2, 1.0, 0)
        rs.append(r)
    a, b = rs
    n = 8 ** 3 * 64
    qs = torch.zeros(n, dtype=torch.int64, device="cuda")
    qc = torch.zeros(n, dtype=torch.int32, device="cuda")
    a.qlearn_bind_sums(qs.data_ptr(), qc.data_ptr())
    for f in range(2):
        a.render(host_api.RT_MODE_PATH, 2 * f, 2)
        b.render(host_api.RT_MODE_PATH, 2 * f, 2)
        sums, cnts = b.qlearn_sums()
        torch.cuda.synchronize()
        assert np.array_equal(qs.cpu().numpy().reshape(sums.shape), sums) and np.array_equal(qc.cpu().numpy().astype(np.uint32).reshape(cnts.shape), cnts)
        assert cnts.sum() > 0
        qs.mul_(2), qc.mul_(2)  # two ranks that rendered the same rows: what the in-place all-reduce leaves
        torch.cuda.synchronize()
        b.qlearn_set_sums(2 * sums, 2 * cnts)
        a.qlearn_apply(), b.qlearn_apply()
        assert np.array_equal(a.qlearn_table().view(np.uint32), b.qlearn_table().view(np.uint32))
        assert int(qs.abs().sum().item()) == 0 and int(qc.abs().sum().item()) == 0  # the apply consumed them, in the caller's arrays
    a.close(), b.close()


def test_qlearning_flag_is_not_the_wide_walks(scenes, host_api):
    """the sampler's overflow word used to be flags[2], which is also the 4-wide occlusion walk's count of rays it handed
    back to the binary walk.  An axis-aligned shadow ray (not 'clean') on a scene with wide nodes left it > 0, and the next
    rt_qlearn_apply / rt_qlearn_get_sums failed with a spurious RT_E_OVERFLOW.  The sampler has its own word now."""
    r = host_api.HostRenderer(64, 40)
    scenes.REGISTRY["mixed_small"](r.scene)  # one scene BVH: the wide walk is on by default
    r.commit()
    r.qlearn_enable(4, (-4, -1, -4), (4, 5, 6), 0.3, 0.2, 1.0, 0)
    O = np.array([[0.0, 3.0, 0.0], [0.5, 3.0, 0.5]], np.float32)
    D = np.array([[0.0, -1.0, 0.0], [0.0, -1.0, 0.0]], np.float32)  # axis-aligned: handed back by the wide walk
    r.is_occluded(O, D)
    r.render(host_api.RT_MODE_PATH, 0, 2)
    r.qlearn_sums()
    r.qlearn_apply()
    r.synchronize()
    r.close()


@pytest.mark.parametrize("name,kw,w,h,box", [("mixed_small", {}, 64, 40, ((-4, -1, -4), (4, 5, 6))), ("pretty_tlas", {"n_instances": 4}, 96, 54, ((-6, -1.5, -1), (8, 5, 10))),
                                             ("tlas_test2", {}, 64, 40, ((-6, -1, -2), (6, 7, 8))),
                                             ("bigb_instanced", {"n": 16, "mesh": "lowBigB"}, 96, 54, ((-12, -2, -8), (12, 10, 16))),  # BASELINE config 5's layout ("Q-learning sampler on") ...
                                             ("bigb_instanced", {"n": 16, "mesh": "BigB"}, 192, 108, ((-12, -2, -8), (12, 10, 16)))])  # ... and its real mesh: BigB.obj x 16, 1/20 size
@pytest.mark.parametrize("mask", [0, 3])
def test_qlearning_sampler(name, kw, w, h, box, mask, scenes, oracle_api, host_api):
    """SURVEY.md 8f N4 / BASELINE config 5 "Q-learning sampler on": Dahm & Keller's guided sampling of the indirect bounce
    (csrc/rt_qlearn.h, rt_qlearn_*).  The reference snapshot holds no code for it (F2), so this is PARITY UNPINNED: the device
    is held against the oracle's statement of the same scheme (oracle/orc_qlearn.h) -- after every batch the pending reward
    sums and counts are equal integer for integer, the learned table bit for bit, the frames within the radiance tolerance --
    and against itself: the rows of a batch rendered as two shards (the sums accumulate, the table is read-only inside a
    batch) give the same frame and the same sums, which is the sharding rule the design states.  The learned table must
    actually differ from its start.  (That the guided estimator is unbiased is held by test_qlearning_sampler_is_unbiased on a
    scene without area lights: here a path that runs into a light's disk is +inf, Q7, and the guided sampler aims at the lights.)"""
    o, orr, r, d = make_pair(scenes.REGISTRY[name], oracle_api, host_api, w, h, **kw)
    orr.scene.set_raytracer(False)
    r.clear(); r.render(host_api.RT_MODE_PATH, 0, 24)
    plain = r.accumulator()[..., :3] / 24
    # mask 3: every fourth sample (by the state of its random stream) pays rewards, all samples pick guided
    orr.qlearn_enable(8, box[0], box[1], 0.3, 0.2, 1.0, mask)
    r.qlearn_enable(8, box[0], box[1], 0.3, 0.2, 1.0, mask)
    orr.clear(); r.clear()
    frames = 3
    for b in range(5):
        orr.render(b * frames, frames, nthreads=0)
        r.render(host_api.RT_MODE_PATH, b * frames, frames)
        so, co, _ = orr.qlearn_state()
        sg, cg = r.qlearn_sums()
        assert co.sum() > 0 and np.array_equal(co, cg), b
        assert np.array_equal(so, sg), b
        ref, got = orr.accumulator(), r.accumulator()
        err, cls_ok = rel_err(got[..., :3], ref[..., :3])
        assert cls_ok and err.max() <= RADIANCE_TOL, (b, err.max())
        orr.qlearn_apply(); r.qlearn_apply()
        assert np.array_equal(orr.qlearn_state()[2].view(np.uint32), r.qlearn_table().view(np.uint32)), b
    # the rows of one batch as two shards, from the same (read-only) table: the same frame, the same pending sums
    zs, zc = np.zeros_like(sg), np.zeros_like(cg)
    r.clear(); r.qlearn_set_sums(zs, zc)
    r.render(host_api.RT_MODE_PATH, 50, frames)
    full, (s_full, c_full) = r.accumulator().copy(), r.qlearn_sums()
    r.clear(); r.qlearn_set_sums(zs, zc)
    r.render_rows(host_api.RT_MODE_PATH, 50, frames, 0, 2, (h + 1) // 2)
    r.render_rows(host_api.RT_MODE_PATH, 50, frames, 1, 2, h // 2)
    s_two, c_two = r.qlearn_sums()
    assert np.array_equal(r.accumulator().view(np.uint32), full.view(np.uint32))
    assert np.array_equal(s_two, s_full) and np.array_equal(c_two, c_full)
    r.qlearn_set_sums(zs, zc)
    tab = r.qlearn_table()
    assert tab.min() < 0.9 and tab.max() > 1.1  # it learned something
    r.qlearn_disable()
    r.clear(); r.render(host_api.RT_MODE_PATH, 0, 24)
    assert np.array_equal((r.accumulator()[..., :3] / 24).view(np.uint32), plain.view(np.uint32))  # off again: the plain sampler's frame
    r.close()


# ---- rt_build_bvh: bvh::Build (binned SAH) on the device, SURVEY.md section 8f N1 ----------------------

class _Recorder:
    """Scene-builder protocol front that forwards to an OracleScene and keeps the spheres and planes."""

    def __init__(self, o):
        self.o, self.spheres, self.planes, self.meshes = o, [], [], []

    def __getattr__(self, name):
        return getattr(self.o, name)

    def sphere(self, idx, mat, pos, r):
        self.spheres.append(list(pos) + [r])
        return self.o.sphere(idx, mat, pos, r)

    def plane(self, idx, mat, N, d):
        self.planes.append(list(N) + [d])
        return self.o.plane(idx, mat, N, d)

    def mesh_obj(self, *a, **k):
        m = self.o.mesh_obj(*a, **k)
        self.meshes.append(m)
        return m

    def mesh_tri(self, *a, **k):
        m = self.o.mesh_tri(*a, **k)
        self.meshes.append(m)
        return m

    def mesh_raw(self, *a, **k):
        m = self.o.mesh_raw(*a, **k)
        self.meshes.append(m)
        return m


def _same_tree(got, ref):
    nodes, prim = got
    assert len(nodes) == ref["nodes_used"]
    assert np.array_equal(prim, ref["prim_idx"])
    keep = np.ones(len(nodes), dtype=bool)
    keep[1] = False  # node 1 is never allocated (Q4)
    assert np.array_equal(nodes[keep], ref["nodes"][:ref["nodes_used"]][keep])


@pytest.mark.parametrize("name,kw", [("mixed_small", {}), ("scene3", {"force_diffuse": False, "split": 0}), ("background", {}), ("tower", {}),
                                     ("mixed_small", {"split": 1}), ("mixed_small", {"split": 2}), ("mixed_small", {"split": 3}),
                                     ("scene3", {"force_diffuse": True, "split": 3}),  # BASELINE config 2's tree: full-sweep SAH
                                     ("scene3", {"force_diffuse": False, "split": 1}), ("background", {"split": 2})])
def test_device_build_equals_reference_build_scene(name, kw, scenes, oracle_api, host_api):
    """Scene bvh over triangles + spheres + planes: the device builder must return the tree the oracle's
    restatement of bvh::Build returns, bit for bit (numbering, boxes, primitiveIdx)."""
    o = oracle_api.OracleScene()
    rec = _Recorder(o)
    scenes.REGISTRY[name](rec, **kw)
    ref = o.bvh_dump(-1)
    tris = [o.mesh_tris(m)[0][:, :9] for m in rec.meshes]
    tv = np.concatenate(tris) if tris else np.zeros((0, 9), np.float32)
    assert len(tv) == ref["NTri"] and len(rec.spheres) == ref["NSph"] and len(rec.planes) == ref["NPla"]
    r = host_api.HostRenderer(8, 8)
    _same_tree(r.build_bvh(tv, rec.spheres or None, rec.planes or None, split=kw.get("split", 0)), ref)
    r.close()


@pytest.mark.parametrize("mesh,split", [("unity", 0), ("BigB", 0), ("lowBigB", 0), ("BigB", 1), ("BigB", 2), ("lowBigB", 3), ("three", 3), ("stellatedDode", 3)])
def test_device_build_equals_reference_build_mesh(mesh, split, scenes, oracle_api, host_api):
    """The BLAS of a mesh (bvh(Mesh*)): 12,584 / 11,830 triangles, ~24 levels; the four split methods of bvh.h:38-43
    (the quadratic full-sweep SAH on the small meshes)."""
    from conftest import pkg
    o = oracle_api.OracleScene()
    if mesh == "unity":
        scenes.REGISTRY["pretty_tlas"](o, n_instances=2, split=split)
    else:
        m = o.mesh_obj(1, pkg("assets").obj_path(mesh), o.diffuse(0.8, (1, 1, 1)), (0, 0.5, 0), 1)
        o.build_tlas(split, [(m, o.trs((0, 0, 3), 4, 0.0, 0.5, 0.0))])
    ref = o.bvh_dump(0)
    tv = o.mesh_tris(0)[0][:, :9]
    r = host_api.HostRenderer(8, 8)
    if not np.isfinite(tv).all():
        with pytest.raises(RuntimeError):
            r.build_bvh(tv, split=split)
    else:
        _same_tree(r.build_bvh(tv, split=split), ref)
    r.close()


@pytest.mark.parametrize("name,kw", [("tlas_test2", {}), ("pretty_tlas", {"n_instances": 8}), ("bigb_instanced", {"n": 16, "mesh": "lowBigB"}), ("bigb_instanced", {"n": 1, "mesh": "lowBigB"})])
def test_device_tlas_build_equals_reference(name, kw, scenes, oracle_api, host_api):
    """tlas::build (tlas.cpp:13-48) on the device: same node order, child packing and boxes as the oracle's restatement."""
    o = oracle_api.OracleScene()
    scenes.REGISTRY[name](o, **kw)
    b6 = np.stack([o.instance_dump(i)["bounds"] for i in range(o.n_instances)])
    r = host_api.HostRenderer(8, 8)
    assert np.array_equal(r.build_tlas(b6), o.tlas_dump())
    r.close()


def test_device_tlas_build_sizes(host_api):
    """1, 2, 3, 100 and 256 instances with random boxes against a plain restatement of the clustering loop; > 256 is refused."""
    r = host_api.HostRenderer(8, 8)
    rng = np.random.default_rng(12)
    def reference(b6):
        n = len(b6)
        node = np.zeros((2 * n + 1, 8), np.uint32)
        nf = node.view(np.float32)
        idx = list(range(1, n + 1))
        for i in range(n):
            nf[1 + i, 0:3], nf[1 + i, 4:7] = b6[i, :3], b6[i, 3:]
            node[1 + i, 7] = i
        used = n + 1
        live = n
        def best(A):
            ids = np.array(idx[:live])
            e = np.maximum(nf[idx[A], 4:7][None], nf[ids, 4:7]) - np.minimum(nf[idx[A], 0:3][None], nf[ids, 0:3])  # float32 throughout
            area = (e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2]) + e[:, 2] * e[:, 0]
            if A < live:
                area[A] = np.inf
            B = int(np.argmin(area))  # the first minimum, like the strict '<' of FindBestMatch
            return B if area[B] < np.float32(1e30) else -1
        A, B = 0, best(0) if n > 1 else 0
        while live > 1:
            Cc = best(B)
            if A != Cc:
                A, B = B, Cc
                continue
            ia, ib = idx[A], idx[B]
            node[used, 3] = ia + (ib << 16)
            nf[used, 0:3] = np.minimum(nf[ia, 0:3], nf[ib, 0:3])
            nf[used, 4:7] = np.maximum(nf[ia, 4:7], nf[ib, 4:7])
            idx[A] = used
            used += 1
            idx[B] = idx[live - 1]  # the list only shrinks logically: A may now sit past its end, as in the reference
            live -= 1
            B = best(A)
        node[0] = node[idx[A]]
        return node[:used]
    for n in (1, 2, 3, 100, 256):
        lo = rng.uniform(-20, 20, (n, 3)).astype(np.float32)
        b6 = np.concatenate([lo, lo + rng.uniform(0.1, 5, (n, 3)).astype(np.float32)], 1)
        assert np.array_equal(r.build_tlas(b6), reference(b6)), n
    with pytest.raises(RuntimeError):
        r.build_tlas(np.zeros((257, 6), np.float32))
    r.close()


def test_device_build_degenerate_inputs(oracle_api, host_api):
    """Equal centroids (no split plane), one / two primitives, all primitives on one side of every plane,
    and the refusals."""
    r = host_api.HostRenderer(8, 8)
    rng = np.random.default_rng(3)
    for tv in (rng.uniform(-1, 1, (1, 9)), rng.uniform(-1, 1, (2, 9)), np.tile(rng.uniform(-1, 1, (1, 9)), (40, 1)),
               np.concatenate([np.tile(rng.uniform(-1, 1, (1, 9)), (30, 1)), rng.uniform(5, 6, (1, 9))]),
               rng.uniform(-3, 3, (5000, 9)), rng.normal(size=(777, 9)) * np.array([10, 0.01, 1] * 3)):
        tv = tv.astype(np.float32)
        o = oracle_api.OracleScene()
        m = o.mesh_raw(1, o.diffuse(0.8, (1, 1, 1)), tv)
        o.build(0)
        _same_tree(r.build_bvh(tv), o.bvh_dump(-1))
    with pytest.raises(RuntimeError):
        r.build_bvh(None, None, [[0, 1, 0, 1]])  # planes only
    bad = rng.uniform(-1, 1, (10, 9)).astype(np.float32)
    bad[3, 4] = np.nan
    with pytest.raises(RuntimeError):
        r.build_bvh(bad)
    r.close()


@pytest.mark.parametrize("name,kw,blas", [("mixed_small", {}, -1), ("tlas_test2", {}, 0), ("pretty_tlas", {"n_instances": 3}, 1),
                                          ("scene3", {"force_diffuse": True, "split": 3}, -1), ("mixed_small", {"split": 1}, -1), ("bigb_instanced", {"n": 16, "mesh": "lowBigB"}, 0)])
def test_host_mirror_builds_on_the_device(name, kw, blas, scenes, oracle_api, host_api):
    """rapt::Scene with deviceBuild set: BuildBVH / BuildTLAS go through bvh::BuildOnDevice -> rt_build_bvh.
    Same arrays as the oracle's builder, and the rendered frame is the host-built scene's frame."""
    o, orr, r, d = make_pair(scenes.REGISTRY[name], oracle_api, host_api, 64, 40, **kw)
    r.render(host_api.RT_MODE_WHITTED, 0, 1)
    want = r.accumulator().copy()
    r2 = host_api.HostRenderer(64, 40)
    r2.scene.device_build(r2.ctx)
    scenes.REGISTRY[name](r2.scene, **kw)
    ref, got = o.bvh_dump(blas), r2.scene.bvh_dump(blas)
    assert got["nodes_used"] == ref["nodes_used"] and np.array_equal(got["prim_idx"], ref["prim_idx"])
    keep = np.ones(ref["nodes_used"], dtype=bool)
    keep[1] = False
    assert np.array_equal(got["nodes"][:ref["nodes_used"]][keep], ref["nodes"][:ref["nodes_used"]][keep])
    assert got["max_depth"] == ref["max_depth"]
    if d["tlas"]:
        assert np.array_equal(r2.scene.tlas_dump(), o.tlas_dump())  # tlas::BuildOnDevice
    r2.commit()
    if "camera" in d:
        c = d["camera"]
        r2.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
    r2.render(host_api.RT_MODE_WHITTED, 0, 1)
    assert np.array_equal(r2.accumulator().view(np.uint32), want.view(np.uint32))
    r2.close()


def test_small_entry_points_check_their_arguments(scenes, host_api):
    """The entry points added in round 6 report bad arguments as errors: rt_set_scene_raytracer (-1, 0, 1 only), rt_device_pci_bus_id (a buffer
    of 16 bytes at least, a device that exists), rt_gather_begin (a context), rt_qlearn_bind_sums (the sampler on)."""
    import ctypes as C
    L = host_api.rt_lib()
    r = host_api.HostRenderer(16, 8)
    scenes.REGISTRY["mixed_small"](r.scene)
    r.commit()
    for bad in (-2, 2, 7):
        assert L.rt_set_scene_raytracer(r.ctx, bad) < 0
    for ok in (0, 1, -1):
        assert L.rt_set_scene_raytracer(r.ctx, ok) == 0
    buf = C.create_string_buffer(64)
    assert L.rt_device_pci_bus_id(0, buf, 8) < 0 and L.rt_device_pci_bus_id(9999, buf, 64) < 0
    assert L.rt_device_pci_bus_id(0, buf, 64) == 0 and host_api.device_pci_bus_id(0) == buf.value.decode() and ":" in buf.value.decode()
    assert L.rt_gather_begin(None) < 0 and L.rt_gather_begin(r.ctx) == 0
    assert L.rt_qlearn_bind_sums(r.ctx, None, None) < 0  # the sampler is off
    r.close()
