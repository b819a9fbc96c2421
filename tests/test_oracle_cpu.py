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
    for seed in (0, 1, 2, 61, 0x12345678, 0xDEADBEEF, 0xFFFFFFFE, 1768515948):
        u = np.zeros(4096, np.uint32)
        f = np.zeros(4096, np.float32)
        L.orc_rng_stream(C.c_uint(seed), 4096, u.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p))
        s = wang(((seed + 1) * 17) & 0xFFFFFFFF) or 0x9E3779B9  # InitSeed (template/template.cpp:680-683), zero state replaced
        for i in range(4096):
            s = xorshift32(s)
            assert u[i] == s
        assert np.array_equal(f, u.astype(np.float32) * np.float32(2.3283064365387e-10))
        assert np.array_equal(u, GOLD["rng_u_%x" % seed])
        assert np.array_equal(f.view(np.uint32), GOLD["rng_f_%x" % seed].view(np.uint32))


def test_hemisphere_fresnel_refract_golden(oracle_api):
    L = oracle_api.lib()
    nrm = GOLD["hemi_normals"]
    out = np.zeros((256, 3), np.float32)
    L.orc_hemisphere(C.c_uint(99), 256, nrm.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out.view(np.uint32), GOLD["hemi_out"].view(np.uint32))
    assert ((out * nrm).sum(1) > 0).all()  # on the normal's side
    assert np.allclose(np.linalg.norm(out, axis=1), 1, atol=1e-6)
    I, N = GOLD["fr_I"], GOLD["fr_N"]
    kr = np.zeros(1024, np.float32)
    rf = np.zeros((1024, 3), np.float32)
    L.orc_fresnel(1024, I.ctypes.data_as(C.c_void_p), N.ctypes.data_as(C.c_void_p), C.c_float(1.5), kr.ctypes.data_as(C.c_void_p))
    L.orc_refract(1024, I.ctypes.data_as(C.c_void_p), N.ctypes.data_as(C.c_void_p), C.c_float(1 / 1.5), rf.ctypes.data_as(C.c_void_p))
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


def _golden_mod():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_goldens_regenerate_bit_for_bit(oracle_api):
    """Drift pin: the generator run on today's oracle reproduces every committed array (per-primitive ray
    vectors, RNG streams, builder / TLAS dumps, primary maps and Whitted / path accumulators of six scenes)."""
    out = _golden_mod().generate()
    assert sorted(out) == sorted(GOLD.files)
    for k in GOLD.files:
        assert out[k].dtype == GOLD[k].dtype and out[k].tobytes() == GOLD[k].tobytes(), k


def test_goldens_do_not_depend_on_the_compiler(oracle_api, tmp_path):
    """The oracle built with ROCm's clang++ instead of g++ (same strict-fp flags) must give the same bits:
    a difference would mean the restatement leans on unspecified behaviour or on one compiler's code generation
    (evaluation order, contraction, libm variants), which the reference's arithmetic must not."""
    import subprocess, sys
    so = oracle_api.build_clang()
    out = str(tmp_path / "clang.npz")
    env = dict(os.environ, ORACLE_LIB=so)
    subprocess.check_call([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden.py"), "--out", out], env=env)
    other = np.load(out)
    assert sorted(other.files) == sorted(GOLD.files)
    for k in GOLD.files:
        assert other[k].tobytes() == GOLD[k].tobytes(), k


def test_reference_probe_observations(scenes, oracle_api):
    """The only numbers in this repository that come from the REFERENCE ITSELF: the survey's probe build of the
    patched reference sources rendered the default scene
    (instantiateBackgroundScene, template/scene.h:791-813) at 600x400 in Whitted mode and observed
      * 9,790 of 240,000 accumulator pixels are +inf (a directly viewed AreaLight, Q7),
      * gprof over two frames (480 k primary rays): 1.04 M bvh::BIntersect calls, 29.3 M Triangle::Intersect calls.
    All three are independent of the (missing) sky texture.  The oracle must reproduce them: the first exactly,
    the gprof figures to the three digits they were recorded with.  A fourth observation of the same probe (SURVEY.md
    8c "Observed"): TLASSceneTest2 (template/scene.h:941-972, BigB.obj x 3 instances through the TLAS) at 600x400 in
    Whitted mode leaves 0 non-finite accumulator pixels (any finite sky gives the same count)."""
    s = oracle_api.OracleScene()
    scenes.background_scene(s)
    s.set_raytracer(True)
    r = oracle_api.OracleRenderer(s, 600, 400)
    cnt = r.render(0, 1, nthreads=0)
    a = r.accumulator()[..., :3]
    assert int(np.isposinf(a).any(-1).sum()) == 9790
    assert not np.isnan(a).any()
    # Whitted frames are deterministic (no RNG), so two frames cost exactly twice one
    assert round(2 * cnt["rays_nearest"] / 1e6, 2) == 1.04
    assert round(2 * cnt["tri_intersect_calls"] / 1e6, 1) == 29.3
    # "~28 triangle tests per traversal" 
    assert round(cnt["tri_intersect_calls"] / cnt["rays_nearest"]) == 28
    r.close(); s.close()
    s = oracle_api.OracleScene()
    scenes.tlas_test2(s, mesh="BigB")
    s.set_raytracer(True)
    r = oracle_api.OracleRenderer(s, 600, 400)
    r.render(0, 1, nthreads=0)
    a = r.accumulator()[..., :3]
    assert int((~np.isfinite(a)).sum()) == 0
    assert (a > 0).any(-1).mean() > 0.9  # and it is a picture, not a black frame
    r.close(); s.close()


def test_qlearning_sampler_cpu(scenes, oracle_api):
    """The oracle's statement of the Q-learning guided sampler (oracle/orc_qlearn.h; the reference holds no code for it, SURVEY
    F2: parity unpinned).  Its rewards are integer sums, so the learned table and the frames must not depend on the number of
    threads that rendered the batch; the table stays positive, its band sums are the sums of its values, rewards are bounded,
    and the guided estimate agrees with the unguided one on average."""
    runs = {}
    for threads in (1, 4):
        s = oracle_api.OracleScene()
        scenes.mixed_small(s)
        s.set_raytracer(False)
        r = oracle_api.OracleRenderer(s, 40, 28)
        r.qlearn_enable(6, (-4, -1, -4), (4, 5, 6), 0.3, 0.2, 1.0)
        for b in range(4):
            r.render(3 * b, 3, nthreads=threads)
            sums, cnts, _ = r.qlearn_state()
            assert cnts.sum() > 0 and sums.min() >= 0 and sums.max() <= 64 * 65536 * int(cnts.max())
            r.qlearn_apply()
        tab = r.qlearn_state()[2]
        runs[threads] = (r.accumulator().copy(), tab.copy())
        assert tab.min() > 0 and tab.min() < 0.9 < 1.1 < tab.max()
        if threads == 1:
            # the V rows a diffuse hit's reward reads: V[cell][m] = sum_p Q[cell][p] max(0, d_m . d_p) over the patch centres, following
            # every update of Q and a loaded table; a patch centre lies in its own patch, as do directions next to the sector boundaries
            v, cen = r.qlearn_v()
            wgt = np.maximum(0.0, cen.astype(np.float64) @ cen.astype(np.float64).T)
            assert np.allclose(v, tab.astype(np.float64) @ wgt.T, rtol=1e-5, atol=1e-6)
            assert [r.qlearn_patch_of(c) for c in cen] == list(range(64))
            for j in range(8):
                for dphi in (0.01, 0.77):
                    phi = (j + 0) * np.pi / 4 + dphi
                    assert r.qlearn_patch_of((0.6 * np.cos(phi), 0.6 * np.sin(phi), -0.8)) == 8 * 0 + j
                    assert r.qlearn_patch_of((0.6 * np.cos(phi), 0.6 * np.sin(phi), 0.8)) == 8 * 7 + j
            assert r.qlearn_patch_of((0, 0, 1)) == 56 and r.qlearn_patch_of((0, 0, -1)) == 0 and r.qlearn_patch_of((float("nan"),) * 3) in range(64)
            r.qlearn_set_table(np.full_like(tab, 2.0))
            v2, _ = r.qlearn_v()
            assert np.allclose(v2, 2.0 * wgt.sum(1)[None, :], rtol=1e-5)
            r.qlearn_set_table(tab)
            assert np.array_equal(r.qlearn_v()[0].view(np.uint32), v.view(np.uint32))
        r.close(); s.close()
    assert np.array_equal(runs[1][1].view(np.uint32), runs[4][1].view(np.uint32))
    assert np.array_equal(runs[1][0].view(np.uint32), runs[4][0].view(np.uint32))
    s = oracle_api.OracleScene()
    scenes.mixed_small(s)
    s.set_raytracer(False)
    r = oracle_api.OracleRenderer(s, 40, 28)
    r.render(0, 12, nthreads=0)
    plain = r.accumulator()[..., :3] / 12
    guided = runs[1][0][..., :3] / 12
    fin = np.isfinite(plain) & np.isfinite(guided)
    assert abs(guided[fin].mean() - plain[fin].mean()) <= 0.1 * plain[fin].mean()
    r.close(); s.close()


def test_qlearning_reward_against_the_exact_integral(scenes, oracle_api):
    """What the diffuse reward gives up: eq. 8's integral over the hemisphere of the hit's normal n is, over the 64
    patch centres, sum_p Q[cell][p] max(0, n . d_p); the sampler reads V[cell][patch(n)] instead -- the same sum with n replaced
    by the centre of the patch it points into (one table read per hit instead of 64 dot products; csrc/rt_qlearn.h,
    oracle/orc_qlearn.h).  The two statements cannot see the difference in each other, so it is measured here: on a table the
    oracle learned and on a synthetic peaky one, for random normals, the quantised reward stays within a few per cent (rms) of
    the exact sum (single normals up to ~50 % off on a learned table, ~75 % on a peaky one) and is unbiased to 2 %.  (numpy on the oracle's own V rows, centres and patch_of.)"""
    s = oracle_api.OracleScene()
    scenes.mixed_small(s)
    s.set_raytracer(False)
    r = oracle_api.OracleRenderer(s, 40, 28)
    r.qlearn_enable(6, (-4, -1, -4), (4, 5, 6), 0.3, 0.2, 1.0)
    for b in range(4):
        r.render(3 * b, 3, nthreads=0)
        r.qlearn_apply()
    rng = np.random.default_rng(5)
    n = rng.normal(size=(1500, 3))
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    m = np.array([r.qlearn_patch_of(tuple(x)) for x in n.astype(np.float32)])
    learned = r.qlearn_state()[2].astype(np.float64)
    peaky = np.full_like(learned, 0.3)
    peaky[np.arange(len(peaky)), rng.integers(0, 64, len(peaky))] = 10.0  # one direction holds nearly all the light
    for name, tab, rms_max, worst_max in (("learned", learned, 0.05, 0.6), ("peaky", peaky, 0.15, 0.9)):
        r.qlearn_set_table(tab.astype(np.float32))
        v, cen = r.qlearn_v()
        cells = rng.integers(0, len(tab), len(n))
        exact = (tab[cells] * np.maximum(0.0, n @ cen.astype(np.float64).T)).sum(1)
        approx = v[cells, m].astype(np.float64)
        rel = (approx - exact) / exact
        assert abs(rel.mean()) <= 0.02, (name, rel.mean())
        assert np.sqrt((rel ** 2).mean()) <= rms_max, (name, np.sqrt((rel ** 2).mean()))
        assert np.abs(rel).max() <= worst_max, (name, np.abs(rel).max())
        assert np.corrcoef(approx, exact)[0, 1] >= 0.9, name
    r.close(); s.close()


def test_primitive_vectors_have_hits():
    """The per-primitive golden vectors are only worth something if they exercise both outcomes."""
    for kind in ("triangle", "sphere", "plane", "disk"):
        obj = GOLD[kind + "_obj_1e-06"]
        assert 0.1 < (obj != -1).mean() < 0.98, kind
        assert 0.02 < GOLD[kind + "_occ"].mean() < 0.98 or kind == "disk", kind
    d = GOLD["aabb_dist"]
    assert 0.1 < (d != np.float32(1e30)).mean() < 0.9


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
