"""CPU tier for the product's host side: the C++ host mirror's loaders and BVH / TLAS builders
against the oracle (bit-exact node arrays), the C ABI library's exports, and loud failure without a
GPU.  No device compute happens here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol(host_api):
    hdr = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    declared = set(re.findall(r"^(?:int|void\*?|rt_ctx\*|const char\*)\s+\*?(rt_[a-z_]+)\s*\(", hdr, re.M))
    assert declared == set(host_api.RT_SYMBOLS), declared ^ set(host_api.RT_SYMBOLS)
    L = host_api.rt_lib()
    for sym in declared:
        assert hasattr(L, sym), sym


def test_no_gpu_fails_loudly(host_api):
    """On a box without a GPU the product refuses to run (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = host_api.rt_lib()
    assert L.rt_create(0, 64, 64) is None
    msg = L.rt_last_error(None).decode()
    assert "no HIP device" in msg or "gfx950" in msg
    with pytest.raises(RuntimeError):
        host_api.HostRenderer(64, 64)


def test_product_does_not_load_the_oracle():
    """The product path (package + bench's GPU leg) must not import, link or load oracle/."""
    pkg_dir = os.path.join(ROOT, "ray-and-pathtracer_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".h", ".cpp", ".hip", ".inc")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "from oracle" not in txt and "import oracle" not in txt and "orc_" not in txt, os.path.join(dirpath, f)


SCENES = [("background", {}), ("scene3", {"force_diffuse": False, "split": 0}), ("scene3", {"force_diffuse": False, "split": 1}),
          ("scene3", {"force_diffuse": False, "split": 2}), ("scene3", {"force_diffuse": False, "split": 3}),
          ("mixed_small", {"split": 0}), ("mixed_small", {"split": 3}), ("tlas_test2", {}), ("tlas_test2", {"mesh": "BigB"}),
          ("pretty_tlas", {"n_instances": 8}), ("tower", {}), ("bigb_instanced", {"n": 16}),
          # the remaining factories of template/scene.h:791-1209 (scenes.REFERENCE_FACTORIES)
          ("pretty_scene1", {}), ("pretty_animation", {}), ("bigb_scene", {}), ("christ_scene", {}), ("tlas_test", {}), ("scene1", {}),
          ("scene2", {}), ("scene4", {}), ("scene5", {}), ("scene6", {}), ("scene7", {"nx": 48, "ny": 48})]


@pytest.mark.parametrize("name,kw", SCENES)
def test_builders_match_oracle(name, kw, scenes, oracle_api, host_api):
    o = oracle_api.OracleScene()
    h = host_api.HostScene()
    d = scenes.REGISTRY[name](o, **kw)
    scenes.REGISTRY[name](h, **kw)
    blas_ids = range(o.blas_count()) if d["tlas"] else [-1]
    if d["tlas"]:
        assert o.blas_count() == h.blas_count()
    for b in blas_ids:
        A, B = o.bvh_dump(b), h.bvh_dump(b)
        assert A["nodes_used"] == B["nodes_used"] and A["N"] == B["N"] and A["max_depth"] == B["max_depth"]
        assert np.array_equal(A["prim_idx"], B["prim_idx"])
        assert np.array_equal(np.delete(A["nodes"], 1, axis=0), np.delete(B["nodes"], 1, axis=0))  # node 1 is never written (Q4)
    if d["tlas"]:
        assert np.array_equal(o.tlas_dump(), h.tlas_dump())
        n_inst = o.n_instances
        for i in range(n_inst):
            a, b = o.instance_dump(i), h.instance_dump(i)
            assert a["blas"] == b["blas"]
            for k in ("T", "invT", "bounds"):
                assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), (i, k)
    assert h.describe()  # flattening to rt_scene_desc works without a device
    o.close(); h.close()


def test_mesh_loaders_match_oracle_and_quirks(tmp_path, oracle_api, host_api):
    """OBJ ('v' / 'f a//n b//n c//n' only) and .tri loaders: same triangles as the oracle, ids
    1000*group + i, scale-then-offset, and the .tri end-of-file quirk (last record twice, Q10)."""
    obj = tmp_path / "m.obj"
    obj.write_text("# comment\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nv 0.25 0.5 -1.5\nusemtl x\nf 1//1 2//1 3//1\nf 2//1 4//1 3//1\n")
    tri = tmp_path / "m.tri"
    tri.write_text("0 0 0 1 0 0 0 1 0\n0 0 1 1 0 1 0 1 1\n999 999 999 999 999 999 999 999 999")
    o, h = oracle_api.OracleScene(), host_api.HostScene()
    for s in (o, h):
        m = s.diffuse(0.8, (1, 1, 1))
        s.mesh_obj(7, str(obj), m, (0.5, -1.0, 2.0), 2.0)
        s.mesh_tri(3, str(tri), m)
    for mesh in (0, 1):
        (ta, ia), (tb, ib) = o.mesh_tris(mesh), h.mesh_tris(mesh)
        assert np.array_equal(ia, ib) and np.array_equal(ta.view(np.uint32), tb.view(np.uint32))
    t, ids = h.mesh_tris(0)
    assert ids.tolist() == [7000, 7001]
    assert np.allclose(t[0, 0:9], [0.5, -1, 2, 2.5, -1, 2, 0.5, 1, 2])
    assert np.allclose(t[0, 9:12], [0, 0, 1])  # N = normalize(cross(e1, e2))
    t, ids = h.mesh_tris(1)
    assert ids.tolist() == [3000, 3001, 3002, 3003]  # 3 records + the duplicated last one
    assert np.array_equal(t[2].view(np.uint32), t[3].view(np.uint32))
    assert np.isnan(t[3, 9:12]).all()  # degenerate sentinel triangle: NaN normal
    with pytest.raises(RuntimeError):
        h.mesh_obj(1, str(tmp_path / "missing.obj"), 0, (0, 0, 0), 1.0)
    o.close(); h.close()


def test_mat4_products_and_inverse(oracle_api, host_api):
    rng = np.random.default_rng(0)
    Lh, Lo = host_api.host_lib(), oracle_api.lib()
    o, h = oracle_api.OracleScene(), host_api.HostScene()
    for _ in range(50):
        t = rng.uniform(-5, 5, 3)
        s, rx, ry, rz = rng.uniform(0.1, 4), *rng.uniform(-3.2, 3.2, 3)
        A, B = o.trs(t, s, rx, ry, rz), h.trs(t, s, rx, ry, rz)
        assert np.array_equal(A.view(np.uint32), B.view(np.uint32))
        ia, ib = np.zeros(16, np.float32), np.zeros(16, np.float32)
        Lo.orc_mat4_inverse(A.ctypes.data_as(C.c_void_p), ia.ctypes.data_as(C.c_void_p))
        Lh.rth_mat4_inverse(B.ctypes.data_as(C.c_void_p), ib.ctypes.data_as(C.c_void_p))
        assert np.array_equal(ia.view(np.uint32), ib.view(np.uint32))
        assert np.allclose(A.reshape(4, 4).astype(np.float64) @ ia.reshape(4, 4).astype(np.float64), np.eye(4), atol=1e-4)
    o.close(); h.close()


def test_every_reference_factory_is_expressed_as_data(scenes):
    """all 16 scene factories of template/scene.h:791-1209 exist as data in scenes.py (two share a
    definition), each buildable by any implementation of the builder protocol and writable as a scene file."""
    assert len(scenes.REFERENCE_FACTORIES) == 16
    assert set(scenes.REFERENCE_FACTORIES.values()) <= set(scenes.REGISTRY)
    assert len(set(scenes.REFERENCE_FACTORIES.values())) == 15


@pytest.mark.parametrize("name,kw", [("background", {}), ("mixed_small", {"split": 3}), ("scene3", {"force_diffuse": False}),
                                     ("tlas_test2", {}), ("pretty_tlas", {"n_instances": 4}), ("tower", {}),
                                     ("tlas_test", {}), ("scene1", {}), ("scene2", {}), ("scene6", {}), ("scene7", {"nx": 20, "ny": 20})])
def test_scene_file_round_trip(name, kw, tmp_path, scenes, host_api):
    """A scene recorded as a 'rapt-scene 1' file and loaded with Scene::LoadFile builds the same
    acceleration structures and flattens to the same primitives as the scene built through the API."""
    from conftest import pkg
    sf = pkg("scene_file")
    path = str(tmp_path / (name + ".rapt"))
    w = sf.SceneWriter(path)
    d = scenes.REGISTRY[name](w, **kw)
    a, b = host_api.HostScene(), host_api.HostScene()
    scenes.REGISTRY[name](a, **kw)
    b.load_file(path)
    for blas in (range(a.blas_count()) if d["tlas"] else [-1]):
        A, B = a.bvh_dump(blas), b.bvh_dump(blas)
        assert np.array_equal(A["prim_idx"], B["prim_idx"])
        assert np.array_equal(np.delete(A["nodes"], 1, axis=0), np.delete(B["nodes"], 1, axis=0))
    if d["tlas"]:
        assert np.array_equal(a.tlas_dump(), b.tlas_dump())
    assert a.n_meshes() == b.n_meshes()
    for m in range(a.n_meshes()):
        (ta, ia), (tb, ib) = a.mesh_tris(m), b.mesh_tris(m)
        assert np.array_equal(ia, ib) and np.array_equal(ta.view(np.uint32), tb.view(np.uint32))
    assert b.describe()
    a.close(); b.close()


def test_scene_file_errors(tmp_path, host_api):
    s = host_api.HostScene()
    bad = tmp_path / "bad.rapt"
    bad.write_text("rapt-scene 1\nmaterial diffuse 0.8 0.8 0.8 1 1 1 0.2 0.8 2 0 0 1\nsphere 1 3 0 0 0 1\nbuild bvh 0\n")
    with pytest.raises(RuntimeError, match="bad.rapt:3: material index"):
        s.load_file(str(bad))
    bad.write_text("not a scene\n")
    with pytest.raises(RuntimeError, match="rapt-scene 1"):
        host_api.HostScene().load_file(str(bad))
    bad.write_text("rapt-scene 1\nplane 0 0 0 1 0 0\n")
    with pytest.raises(RuntimeError):
        host_api.HostScene().load_file(str(bad))
    s.close()


def test_partition_closed_form_equals_the_loop():
    """csrc/rt_build.h replaces the reference's in-place partition loop (bvh.cpp:296-313) by a closed form
    for where every element ends up; this checks that form against the loop itself on random arrays."""
    import random

    def loop(a, is_left):
        a = list(a)
        i, j = 0, len(a) - 1
        while i <= j:
            if is_left[a[i]]:
                i += 1
            else:
                a[i], a[j] = a[j], a[i]
                j -= 1
        return a, i

    def closed_form(a, is_left):
        n = len(a)
        L = [is_left[x] for x in a]
        n_l = sum(L)
        holes = [p for p in range(n_l) if not L[p]]
        fillers = [q for q in range(n - 1, n_l - 1, -1) if L[q]]
        k_total = len(holes)
        assert k_total == len(fillers)
        f_last = fillers[-1] if k_total else n
        out = [None] * n
        seen_h = seen_f = 0
        for i in range(n):
            if i < n_l:
                if L[i]:
                    d = i
                else:
                    d = (n if seen_h == 0 else fillers[seen_h - 1]) - 1
                    seen_h += 1
            elif L[i]:
                d = holes[k_total - 1 - seen_f]
                seen_f += 1
            else:
                d = i - 1 if i > f_last else (f_last - 1 if i == n_l else i - 1)
            assert out[d] is None
            out[d] = a[i]
        return out, n_l

    rnd = random.Random(11)
    for _ in range(20000):
        n = rnd.randint(0, 40)
        a = list(range(n))
        rnd.shuffle(a)
        p = rnd.random()
        is_left = {x: rnd.random() < p for x in a}
        assert loop(a, is_left) == closed_form(a, is_left)


def test_bench_plumbing_without_a_gpu(tmp_path, monkeypatch):
    """bench.py's host-side plumbing that needs no device: the kernel hash covers the kernel sources and the build string (a profile
    measured on other sources is never quoted: roofline_pmc.json, the out-of-cache files), the frame comparison of parity_check
    (tests/conftest.py rel_err's rule), the CPU allowance parser, and -- `python bench.py --gpus N` with no WORLD_SIZE -- the
    self-start: a child torch.distributed.run with N ranks on 127.0.0.1, before torch is imported, the exit code relayed.  And the
    staleness rule of host_api.build(): the .inc step fragments are the hottest code of the library and count as sources."""
    import importlib, subprocess, sys
    from conftest import ROOT
    bench = importlib.import_module("bench")
    h0 = bench.kernel_hash("flags=[] A=1")
    assert h0 == bench.kernel_hash("flags=[] A=1") and h0 != bench.kernel_hash("flags=[-DX] A=1") and len(h0) == 16
    # the committed counter file and the out-of-cache files are stamped with hashes of this shape
    import json
    pj = json.load(open(os.path.join(ROOT, "profiles", "roofline_pmc.json")))
    assert all(re.fullmatch(r"[0-9a-f]{16}", w["kernel_hash"]) for w in pj["by_world"].values())
    a = np.array([[1.0, 2.0, np.inf], [0.0, 1e-9, np.nan]])
    b = np.array([[1.0, 2.0002, np.inf], [0.0, 0.0, np.nan]])
    err, cls_ok = bench.frame_error(a, b)
    assert cls_ok and abs(err - 0.0002 / 2.0002) < 1e-9
    assert not bench.frame_error(np.array([1.0, np.inf]), np.array([1.0, -np.inf]))[1]
    allowed = bench.cpu_allowance()
    assert allowed is None or allowed > 0
    # an edited step fragment makes the library stale: build() must call make for it
    ha = importlib.import_module("ray-and-pathtracer_amd.host_api")
    calls = []
    monkeypatch.setattr(ha.subprocess, "check_call", lambda cmd, **kw: calls.append(cmd))
    inc = os.path.join(ROOT, "ray-and-pathtracer_amd", "csrc", "rt_step_pair.inc")
    st = os.stat(inc)
    try:
        ha.build()
        assert not any("csrc" in " ".join(c) for c in calls), calls  # fresh: nothing to do
        os.utime(inc, (st.st_atime, os.path.getmtime(ha.RT_SO) + 10))
        ha.build()
        assert any("csrc" in " ".join(c) for c in calls), calls
    finally:
        os.utime(inc, (st.st_atime, st.st_mtime))
    monkeypatch.undo()
    # the self-start, with a stand-in for torch.distributed.run on the path: it records its arguments and exits 7
    fake = tmp_path / "torch" / "distributed"
    fake.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (fake / "__init__.py").write_text("")
    (fake / "run.py").write_text("import sys, json\njson.dump(sys.argv[1:], open(%r, 'w'))\nsys.exit(7)\n" % str(tmp_path / "argv.json"))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PYTHONPATH"] = str(tmp_path) + os.pathsep + env.get("PYTHONPATH", "")
    rc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "2"], env=env, cwd=ROOT).returncode
    assert rc == 7
    argv = json.load(open(tmp_path / "argv.json"))
    # one node, N ranks, the rendezvous on 127.0.0.1 with a port the launcher picks itself (no bind-then-close race: ADVICE r5)
    assert argv[:2] == ["--nnodes=1", "--nproc-per-node"] and argv[2] == "3" and "--rdzv-endpoint=127.0.0.1:0" in argv and argv[argv.index("--local-addr") + 1] == "127.0.0.1"
    assert argv[-4:] == ["--gpus", "3", "--steps", "2"] and argv[-5].endswith("bench.py")


def test_committed_counter_files_are_of_these_kernel_sources():
    """bench.py quotes profiles/roofline_pmc.json and profiles/out_of_cache_pmc.json only while they were measured on the tree's kernel sources
    (kernel_hash over csrc/*.{h,hip,inc}, the Makefile and the library's build / tuning string).  The build string of the committed bench
    line (profiles/final_bench.json) is the one a default context reports, so the check can be made without a GPU: a kernel edit that was not
    followed by the counter passes fails HERE, not silently as a null roofline.frac in the driver's run.  And the roofline block recomputes from
    the files: TA busy fraction = TA busy cycles per launch / the launch time it is given."""
    import importlib, json, argparse
    from conftest import ROOT
    bench = importlib.import_module("bench")
    line = json.loads(open(os.path.join(ROOT, "profiles", "final_bench.json")).read().strip().splitlines()[-1])
    build = line["roofline"]["build"]
    pj = json.load(open(os.path.join(ROOT, "profiles", "roofline_pmc.json")))["by_world"]
    assert {w["kernel_hash"] for w in pj.values()} == {bench.kernel_hash(build)}, "kernel sources changed since profiles/roofline_pmc.json was measured: run profiles/roofline_passes.sh"
    ooc = json.load(open(os.path.join(ROOT, "profiles", "out_of_cache_pmc.json")))
    assert ooc["kernel_hash"] == bench.kernel_hash(build.split(" | ")[0]), "kernel sources changed since profiles/out_of_cache_pmc.json was measured: run profiles/out_of_cache.sh 2048 16 r06"
    assert line["roofline"]["pmc"]["used"] and line["roofline"]["pmc"]["kernel_hash"] == bench.kernel_hash(build)
    # the block, recomputed from the committed files and the line's own inputs
    ha = importlib.import_module("ray-and-pathtracer_amd.host_api")
    work = line["roofline"]["algorithmic_work_per_step"]
    near = dict(work, rays_occluded=0, brute_tests=0, light_tests=0)
    occl = dict(inner_visits=0, prim_tests=0, tlas_inner=0, instance_visits=0, rays_nearest=0, rays_occluded=line["rays_per_step"]["occluded"], brute_tests=0, light_tests=0)
    args = argparse.Namespace(qlearn=0, workload="config3", steps=line["steps"])
    rb = bench.roofline_block(args, ha, near, occl, line["roofline"]["avg_launch_ms"], line["roofline"]["launches_per_step"], 1920, 1080, 64, 1,
                              line["roofline"]["kernel_ms_per_step"], build, line["ms_per_step"] * 1e-3, dict(launches=100, ms=150.0))
    assert rb["bound"] == line["roofline"]["bound"] and rb["pmc"]["used"]
    assert abs(rb["frac"] - line["roofline"]["frac"]) < 1e-3 and 0.5 < rb["frac"] <= 1.0
    k = pj["1"]["kernels"]["k_extend"]
    assert abs(rb["sides"]["ta_busy"]["live"] - k["ta_busy_avg"] * (k["ms"] / k["launches"]) / line["roofline"]["avg_launch_ms"]) < 1e-3
    assert rb["sides"]["hbm_algorithmic_demand_over_peak"] > 1 and rb["sides"]["hbm_counter_frac_of_peak"] < 0.2  # demand served by the caches; what HBM really moves
    assert rb["hbm"]["algorithmic_bytes_per_launch"] == line["roofline"]["hbm"]["algorithmic_bytes_per_launch"]
