"""CPU tier: the oracle against (a) independent restatements of the published algorithms it uses
(Marsaglia xorshift32, Wang hash -- pure Python, written from the algorithm descriptions), (b) the
committed golden vectors (tests/golden/oracle_vectors.npz, drift pin), (c) analytic cases.

The reference ships no tests or fixtures: parity of the oracle with the reference
itself is unpinned, see oracle/README.md."""
import ctypes as C
import os

import numpy as np
import pytest

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.npz"))


def xorshift32(s):
    s ^= (s << 13) & 0xFFFFFFFF
    s ^= s >> 17
    s ^= (s << 5) & 0xFFFFFFFF
    return s & 0xFFFFFFFF


def wang(s):
    s = ((s ^ 61) ^ (s >> 16)) & 0xFFFFFFFF
    s = (s * 9) & 0xFFFFFFFF
    s ^= s >> 4
    s = (s * 0x27D4EB2D) & 0xFFFFFFFF
    s ^= s >> 15
    return s


def test_rng_against_published_algorithms(oracle_api):
    L = oracle_api.lib()
    # Marsaglia (2003) "Xorshift RNGs", triple (13, 17, 5): from state 2463534242 the next state is 723471715
    assert xorshift32(2463534242) == 723471715
    for seed in (0, 1, 0x12345678, 0xFFFFFFFE):
        u = np.zeros(64, np.uint32)
        f = np.zeros(64, np.float32)
        L.orc_rng_stream(C.c_uint(seed), 64, u.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p))
        s = wang(((seed + 1) * 17) & 0xFFFFFFFF) or 0x9E3779B9  # InitSeed (template/template.cpp:680-683), zero state replaced
        for i in range(64):
            s = xorshift32(s)
            assert u[i] == s
            assert f[i] == np.float32(s) * np.float32(2.3283064365387e-10)
        assert np.array_equal(u, GOLD["rng_u_%x" % seed])
        assert np.array_equal(f.view(np.uint32), GOLD["rng_f_%x" % seed].view(np.uint32))


def test_hemisphere_fresnel_refract_golden(oracle_api):
    L = oracle_api.lib()
    nrm = GOLD["hemi_normals"]
    out = np.zeros((64, 3), np.float32)
    L.orc_hemisphere(C.c_uint(99), 64, nrm.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out.view(np.uint32), GOLD["hemi_out"].view(np.uint32))
    assert ((out * nrm).sum(1) > 0).all()  # on the normal's side
    assert np.allclose(np.linalg.norm(out, axis=1), 1, atol=1e-6)
    I, N = GOLD["fr_I"], GOLD["fr_N"]
    kr = np.zeros(128, np.float32)
    rf = np.zeros((128, 3), np.float32)
    L.orc_fresnel(128, I.ctypes.data_as(C.c_void_p), N.ctypes.data_as(C.c_void_p), C.c_float(1.5), kr.ctypes.data_as(C.c_void_p))
    L.orc_refract(128, I.ctypes.data_as(C.c_void_p), N.ctypes.data_as(C.c_void_p), C.c_float(1 / 1.5), rf.ctypes.data_as(C.c_void_p))
    assert np.array_equal(kr.view(np.uint32), GOLD["fr_kr"].view(np.uint32))
    assert np.array_equal(rf.view(np.uint32), GOLD["fr_refract"].view(np.uint32))
    assert ((kr >= 0) & (kr <= 1)).all()
    # normal incidence on glass: ((1.5-1)/(1.5+1))^2 = 0.04
    one = np.zeros(1, np.float32)
    n = np.array([[0, 0, 1]], np.float32)
    i = np.array([[0, 0, -1]], np.float32)
    L.orc_fresnel(1, i.ctypes.data_as(C.c_void_p), n.ctypes.data_as(C.c_void_p), C.c_float(1.5), one.ctypes.data_as(C.c_void_p))
    assert abs(one[0] - 0.04) < 1e-6


def test_aabb_slab_cases(oracle_api):
    L = oracle_api.lib()
    f3 = lambda v: (C.c_float * 3)(*v)
    hit = L.orc_intersect_aabb(f3((0, 0, -5)), f3((0, 0, 1)), C.c_float(1e34), f3((-1, -1, -1)), f3((1, 1, 1)))
    assert hit == 4.0
    assert L.orc_intersect_aabb(f3((0, 0, -5)), f3((0, 0, 1)), C.c_float(3.0), f3((-1, -1, -1)), f3((1, 1, 1))) == np.float32(1e30)  # beyond ray.t
    assert L.orc_intersect_aabb(f3((0, 0, 5)), f3((0, 0, 1)), C.c_float(1e34), f3((-1, -1, -1)), f3((1, 1, 1))) == np.float32(1e30)  # behind
    # a ray running exactly along a box face: 0 * inf = NaN in that slab, and the std::min / std::max
    # ternaries (bvh.cpp:822) then reject the box on either face
    for x in (1.0, -1.0):
        v = L.orc_intersect_aabb(f3((x, 0, -5)), f3((0, 0, 1)), C.c_float(1e34), f3((-1, -1, -1)), f3((1, 1, 1)))
        assert v == np.float32(1e30)


@pytest.mark.parametrize("name,kw", [("background", {}), ("mixed_small", {}), ("scene3", {"force_diffuse": False}), ("tlas_test2", {})])
def test_scene_goldens(name, kw, scenes, oracle_api):
    s = oracle_api.OracleScene()
    d = scenes.REGISTRY[name](s, **kw)
    b = s.bvh_dump(0 if d["tlas"] else -1)
    assert np.array_equal(np.delete(b["nodes"], 1, axis=0), GOLD[name + "_nodes"])
    assert np.array_equal(b["prim_idx"], GOLD[name + "_prim_idx"])
    if d["tlas"]:
        assert np.array_equal(s.tlas_dump(), GOLD[name + "_tlas"])
    r = oracle_api.OracleRenderer(s, 48, 32)
    obj, t, cnt = r.primary_hits(1e-6)
    assert np.array_equal(obj, GOLD[name + "_obj"])
    assert np.array_equal(t.view(np.uint32), GOLD[name + "_t"].view(np.uint32))
    assert [cnt[k] for k in oracle_api.COUNTER_NAMES] == GOLD[name + "_cnt"].tolist()
    s.set_raytracer(True)
    r.render(0, 1)
    assert np.array_equal(r.accumulator().view(np.uint32), GOLD[name + "_whitted"].view(np.uint32))
    s.set_raytracer(False)
    r.clear()
    r.render(0, 4, nthreads=0)  # OpenMP over scanlines; per-pixel RNG streams make it thread-count independent
    assert np.array_equal(r.accumulator().view(np.uint32), GOLD[name + "_path4"].view(np.uint32))
    r.close(); s.close()


def test_bvh_invariants(scenes, oracle_api):
    """Structural properties of the restated builder: every primitive in exactly one leaf, parents
    bound their children (bvh::Refit), node 1 never used, plane leaf split off first."""
    for split in range(4):
        s = oracle_api.OracleScene()
        scenes.mixed_small(s, split=split)
        b = s.bvh_dump(-1)
        nodes = b["nodes"]
        fl = nodes.view(np.float32)
        seen = np.zeros(b["N"], int)
        for i in range(b["nodes_used"]):
            if i == 1:
                continue
            left_first, count = nodes[i, 3], nodes[i, 7]
            if count > 0:
                seen[b["prim_idx"][left_first:left_first + count]] += 1
            else:
                for c in (left_first, left_first + 1):
                    assert (fl[c, 0:3] >= fl[i, 0:3]).all() and (fl[c, 4:7] <= fl[i, 4:7]).all()
        assert (seen == 1).all()
        root = nodes[0]
        assert root[7] == 0 and nodes[root[3] + 1, 7] == b["NPla"]  # right child of the root = the planes
        s.close()


def test_whitted_image_statistics(scenes, oracle_api):
    """Sanity of the integrator on the reference's default scene: the sky shows, the floor is lit,
    the directly viewed light is +inf, nothing is NaN."""
    s = oracle_api.OracleScene()
    scenes.background_scene(s)
    r = oracle_api.OracleRenderer(s, 96, 64)
    r.render(0, 1)
    a = r.accumulator()[..., :3]
    assert not np.isnan(a).any()
    assert np.isposinf(a).any()
    fin = a[np.isfinite(a).all(-1)]
    assert fin.min() >= 0 and 0.05 < fin.mean() < 5
    r.close(); s.close()


def test_zero_hash_stream_does_not_hang(scenes, oracle_api):
    """WangHash(61) == 0, so the stream with index 1768515948 would start xorshift32 in its fixed
    point 0 and RandomVectorInUnitSphere would spin forever; per-sample streams replace that state."""
    assert wang(61) == 0 and ((1768515948 + 1) * 17) & 0xFFFFFFFF == 61
    s = oracle_api.OracleScene()
    scenes.mixed_small(s)
    s.set_raytracer(False)
    r = oracle_api.OracleRenderer(s, 16, 8)
    r.render(0, 1, seed_base=1768515948 - 40)  # pixel 40 of frame 0 gets the zero-hash index
    assert np.isfinite(r.accumulator()[..., 3]).all()
    r.close(); s.close()
