"""The HIP path (through the C ABI) against the COMMITTED golden vectors (tests/golden/oracle_vectors.npz):
fixed expected values that were produced by the oracle once (tests/golden/make_golden.py), reproduce under two
compilers (tests/test_oracle_cpu.py) and do not need the oracle at test time.  Bar as everywhere: ids, t,
normals, occlusion flags bit-exact; radiance within 1e-4 relative, non-finite pixels by class."""
import os
import sys

import numpy as np
import pytest

from conftest import rel_err

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import fixtures as fx  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.npz"))


@pytest.mark.parametrize("kind", ["triangle", "sphere", "plane", "disk"])
def test_primitive_vectors(kind, host_api):
    """2,048 rays against one triangle / sphere / plane / disk light: (t bits, objIdx, normal bits) for both t_min
    values the integrators use, and the any-hit flag with a bounded tmax."""
    r = host_api.HostRenderer(8, 8)
    fx.single_primitive(r.scene, kind)
    r.commit()
    O, D = GOLD[kind + "_O"], GOLD[kind + "_D"]
    for t_min in (1e-6, 0.001):
        got = r.find_nearest(O, D, t_min=t_min)
        assert np.array_equal(got["obj"], GOLD["%s_obj_%g" % (kind, t_min)])
        assert np.array_equal(got["t"].view(np.uint32), GOLD["%s_t_%g" % (kind, t_min)].view(np.uint32))
        hit = got["obj"] != -1
        assert np.array_equal(got["normal"][hit].view(np.uint32), GOLD["%s_n_%g" % (kind, t_min)][hit].view(np.uint32))
    assert np.array_equal(r.is_occluded(O, D, GOLD[kind + "_tmax"]), GOLD[kind + "_occ"])
    r.close()


@pytest.mark.parametrize("name,kw", fx.FIXTURE_SCENES)
def test_scene_vectors(name, kw, scenes, host_api):
    """Per fixture scene at 64x64: the host builders' node arrays, the primary objIdx / t maps with the reference's
    work tallies, and the Whitted (1 frame) and path (1, 4, 16 frames) accumulators."""
    key = fx.scene_key(name, kw)
    r = host_api.HostRenderer(64, 64)
    d = scenes.REGISTRY[name](r.scene, **kw)
    r.commit()
    if "camera" in d:
        c = d["camera"]
        r.set_camera(c["cam_pos"], c["top_left"], c["top_right"], c["bottom_left"])
    b = r.scene.bvh_dump(0 if d["tlas"] else -1)
    assert np.array_equal(np.delete(b["nodes"][:b["nodes_used"]], 1, axis=0), GOLD[key + "_nodes"])
    assert np.array_equal(b["prim_idx"], GOLD[key + "_prim_idx"])
    if d["tlas"]:
        assert np.array_equal(r.scene.tlas_dump(), GOLD[key + "_tlas"])
    r.set_counting(True)
    r.counters()
    obj, t = r.primary_hits(1e-6)
    cnt = r.counters()
    r.set_counting(False)
    assert np.array_equal(obj, GOLD[key + "_obj"])
    assert np.array_equal(t.view(np.uint32), GOLD[key + "_t"].view(np.uint32))
    assert [cnt[k] for k in host_api.COUNTER_NAMES] == GOLD[key + "_cnt"].tolist()
    for mode, frames, gk in ((host_api.RT_MODE_WHITTED, 1, "_whitted"), (host_api.RT_MODE_PATH, 1, "_path1"),
                             (host_api.RT_MODE_PATH, 4, "_path4"), (host_api.RT_MODE_PATH, 16, "_path16")):
        r.clear()
        r.render(mode, 0, frames)
        err, cls_ok = rel_err(r.accumulator()[..., :3], GOLD[key + gk][..., :3])
        assert cls_ok, "non-finite pixels differ by class"
        assert err.max() <= 1e-4, (gk, err.max())
    r.close()
