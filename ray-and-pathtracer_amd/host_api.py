"""ctypes bindings of the product: the C ABI of librt_amd.so (include/rt_amd.h) and the C++ host
mirror librapt_host.so (host/rapt.h).  No CPU fallback exists: loading fails loudly when a library
is missing, and every device call raises when there is no gfx950 GPU.

HostScene implements the scene-builder protocol used by scenes.py (the same protocol the oracle
implements in oracle/oracle_api.py, which this module never imports).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
RT_SO = os.path.join(_HERE, "csrc", "librt_amd.so")
HOST_SO = os.path.join(_HERE, "host", "librapt_host.so")

RT_MODE_WHITTED, RT_MODE_PATH = 0, 1
RT_COUNT_OFF, RT_COUNT_REFERENCE, RT_COUNT_EXECUTED = 0, 1, 2
COUNTER_NAMES = ["inner_visits", "prim_tests", "tlas_inner", "instance_visits",
                 "rays_nearest", "rays_occluded", "brute_tests", "light_tests"]

# every symbol include/rt_amd.h declares
RT_SYMBOLS = ["rt_device_count", "rt_create", "rt_destroy", "rt_last_error", "rt_upload_scene", "rt_set_camera", "rt_set_time",
              "rt_render", "rt_render_rows", "rt_clear", "rt_download_accumulator", "rt_resolve", "rt_accumulator_device_ptr",
              "rt_bind_accumulator", "rt_intersect_batch", "rt_occluded_batch", "rt_primary_hits", "rt_trace_batch",
              "rt_set_counting", "rt_get_counters", "rt_get_counters_split", "rt_set_profiling", "rt_get_profile", "rt_synchronize",
              "rt_build_bvh", "rt_build_bvh_split", "rt_build_tlas", "rt_gather_rows", "rt_gather_begin", "rt_device_of",
              "rt_intersect_scope", "rt_occluded_scope", "rt_sky_color_batch", "rt_trace_batch_energy", "rt_build_info", "rt_tuning_info",
              "rt_qlearn_enable", "rt_qlearn_apply", "rt_qlearn_get_sums", "rt_qlearn_set_sums", "rt_qlearn_get_table", "rt_qlearn_bind_sums",
              "rt_device_pci_bus_id", "rt_set_scene_raytracer"]


class RtQlearnParams(C.Structure):
    _fields_ = [("grid", C.c_int32), ("lo", C.c_float * 3), ("hi", C.c_float * 3), ("alpha", C.c_float), ("epsilon", C.c_float), ("q_init", C.c_float), ("learn_mask", C.c_uint32)]


class RtCamera(C.Structure):
    _fields_ = [("cam_pos", C.c_float * 3), ("top_left", C.c_float * 3), ("top_right", C.c_float * 3),
                ("bottom_left", C.c_float * 3), ("fisheye", C.c_int32), ("view_angle", C.c_float), ("y_angle", C.c_float)]


class RtKernelTime(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("ms", C.c_double)]


class RtProfile(C.Structure):
    _fields_ = [("generate", RtKernelTime), ("extend", RtKernelTime), ("shade", RtKernelTime),
                ("connect", RtKernelTime), ("query", RtKernelTime)]


RT_TRIANGLE_DTYPE = np.dtype([("v0", np.float32, 3), ("v1", np.float32, 3), ("v2", np.float32, 3), ("N", np.float32, 3), ("obj_idx", np.int32), ("material", np.int32)])
RT_SPHERE_DTYPE = np.dtype([("pos", np.float32, 3), ("r2", np.float32), ("invr", np.float32), ("r", np.float32), ("obj_idx", np.int32), ("material", np.int32)])
RT_PLANE_DTYPE = np.dtype([("N", np.float32, 3), ("d", np.float32), ("obj_idx", np.int32), ("material", np.int32)])
RT_BVH_NODE_DTYPE = np.dtype([("aabb_min", np.float32, 3), ("left_first", np.uint32), ("aabb_max", np.float32, 3), ("prim_count", np.uint32)])
RT_HIT_DTYPE = np.dtype([("t", np.float32), ("obj_idx", np.int32), ("material", np.int32), ("normal", np.float32, 3)])


def build(force=False):
    """Compile both product libraries in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    def stale(so, d):
        srcs = [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".h", ".cpp", ".hip", ".inc")) or f == "Makefile"]
        srcs.append(os.path.join(_HERE, "..", "include", "rt_amd.h"))
        return force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale(RT_SO, os.path.join(_HERE, "csrc")):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-s"])
    if stale(HOST_SO, os.path.join(_HERE, "host")) or os.path.getmtime(RT_SO) > os.path.getmtime(HOST_SO):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "host"), "-s"])


_rt = None
_host = None


def rt_lib():
    global _rt
    if _rt is None:
        if not os.path.exists(RT_SO):
            raise RuntimeError("librt_amd.so is not built (run __graft_entry__.build()); there is no fallback path")
        L = C.CDLL(RT_SO, mode=C.RTLD_GLOBAL)
        L.rt_create.restype = C.c_void_p
        L.rt_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.rt_last_error.restype = C.c_char_p
        L.rt_last_error.argtypes = [C.c_void_p]
        L.rt_qlearn_enable.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_qlearn_apply.argtypes = [C.c_void_p]
        L.rt_qlearn_get_sums.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_qlearn_set_sums.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_qlearn_get_table.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_qlearn_bind_sums.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_int]
        L.rt_gather_begin.argtypes = [C.c_void_p]
        L.rt_set_scene_raytracer.argtypes = [C.c_void_p, C.c_int]
        L.rt_build_info.restype = C.c_char_p
        L.rt_build_info.argtypes = []
        L.rt_tuning_info.restype = C.c_char_p
        L.rt_tuning_info.argtypes = [C.c_void_p]
        L.rt_accumulator_device_ptr.restype = C.c_void_p
        L.rt_accumulator_device_ptr.argtypes = [C.c_void_p]
        for name in ["rt_destroy", "rt_upload_scene", "rt_set_camera", "rt_set_time", "rt_clear", "rt_synchronize"]:
            getattr(L, name).argtypes = [C.c_void_p] + ([C.c_void_p] if name in ("rt_upload_scene", "rt_set_camera") else [])
        L.rt_set_time.argtypes = [C.c_void_p, C.c_float]
        L.rt_render.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int]
        L.rt_render_rows.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int]
        L.rt_get_counters_split.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.rt_download_accumulator.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.rt_resolve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.rt_bind_accumulator.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_intersect_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.rt_occluded_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_primary_hits.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        L.rt_trace_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
        L.rt_trace_batch_energy.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
        L.rt_intersect_scope.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.rt_occluded_scope.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_sky_color_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.rt_set_counting.argtypes = [C.c_void_p, C.c_int]
        L.rt_build_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_build_bvh_split.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_build_tlas.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.rt_get_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.rt_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.rt_get_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _rt = L
    return _rt


def host_lib():
    global _host
    if _host is None:
        rt_lib()
        if not os.path.exists(HOST_SO):
            raise RuntimeError("librapt_host.so is not built (run __graft_entry__.build())")
        L = C.CDLL(HOST_SO)
        for name in ["rth_scene_new", "rth_renderer_new", "rth_renderer_new_multi", "rth_renderer_scene", "rth_renderer_ctx", "rth_describe",
                     "rth_renderer_accumulator", "rth_renderer_pixels", "rth_get_sky"]:
            getattr(L, name).restype = C.c_void_p
        L.rth_last_error.restype = C.c_char_p
        L.rth_renderer_error.restype = C.c_char_p
        _host = L
    return _host


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class HostScene:
    """rapt::Scene through the rth_* C entry points.  Standalone (CPU only: loaders and builders) or
    the scene member of a HostRenderer."""

    def __init__(self, handle=None):
        self.L = host_lib()
        self.owned = handle is None
        self.h = C.c_void_p(self.L.rth_scene_new()) if handle is None else C.c_void_p(handle)

    def close(self):
        if self.h and self.owned:
            self.L.rth_scene_free(self.h)
        self.h = None

    def _chk(self, rc):
        if rc < 0:
            raise RuntimeError(self.L.rth_last_error(self.h).decode())
        return rc

    # ---- builder protocol (see scenes.py) ----
    def diffuse(self, albedo, col, ks=0.2, kd=0.8, n=2, emission=0.0, shininess=0.0, rt=True):
        a = albedo if hasattr(albedo, "__len__") else (albedo,) * 3
        return self.L.rth_add_diffuse(self.h, _f3(a), _f3(col), C.c_float(ks), C.c_float(kd), int(n),
                                      C.c_float(emission), C.c_float(shininess), int(rt))

    def metal(self, fuzzy, col, rt=True):
        return self.L.rth_add_metal(self.h, C.c_float(fuzzy), _f3(col), int(rt))

    def glass(self, ir, col, absorption=(0, 0, 0), rt=True):
        return self.L.rth_add_glass(self.h, C.c_float(ir), _f3(col), _f3(absorption), int(rt))

    def area_light(self, idx, pos, strength, col, radius, normal):
        return self.L.rth_add_area_light(self.h, idx, _f3(pos), C.c_float(strength), _f3(col), C.c_float(radius), _f3(normal))

    def dir_light(self, idx, pos, strength, col, normal, r):
        return self.L.rth_add_dir_light(self.h, idx, _f3(pos), C.c_float(strength), _f3(col), _f3(normal), C.c_float(r))

    def sphere(self, idx, mat, pos, r):
        return self.L.rth_add_sphere(self.h, idx, mat, _f3(pos), C.c_float(r))

    def plane(self, idx, mat, N, d):
        return self.L.rth_add_plane(self.h, idx, mat, _f3(N), C.c_float(d))

    def mesh_raw(self, group, mat, v9):
        v9 = np.ascontiguousarray(v9, dtype=np.float32).reshape(-1, 9)
        return self.L.rth_add_mesh_raw(self.h, group, mat, _p(v9), len(v9))

    def mesh_obj(self, group, path, mat, pos, scale):
        return self._chk(self.L.rth_add_mesh_obj(self.h, group, path.encode(), mat, _f3(pos), C.c_float(scale)))

    def mesh_tri(self, group, path, mat):
        return self._chk(self.L.rth_add_mesh_tri(self.h, group, path.encode(), mat))

    def sky(self, pixels):
        px = np.ascontiguousarray(pixels, dtype=np.uint8)
        hgt, w, n = px.shape
        self.L.rth_set_sky(self.h, w, hgt, n, _p(px))

    def load_file(self, path):
        """Scene::LoadFile: build the scene a 'rapt-scene 1' description file names."""
        self._chk(self.L.rth_scene_load_file(self.h, path.encode()))

    def sky_hdr(self, path):
        """Scene's stbi_load(path, ..., 3) of a Radiance .hdr file; returns the 8-bit texels [H, W, 3]."""
        self._chk(self.L.rth_load_sky_hdr(self.h, path.encode()))
        dims = (C.c_int * 3)()
        p = self.L.rth_get_sky(self.h, dims)
        buf = C.cast(p, C.POINTER(C.c_ubyte))
        return np.ctypeslib.as_array(buf, shape=(dims[1], dims[0], dims[2])).copy()

    def trs(self, t, s, rx, ry, rz):
        out = np.zeros(16, dtype=np.float32)
        self.L.rth_mat4_trs(_f3(t), C.c_float(s), C.c_float(rx), C.c_float(ry), C.c_float(rz), _p(out))
        return out

    def device_build(self, ctx):
        """Build BINNEDSAH trees with rt_build_bvh on the device of 'ctx' (None: on the host again)."""
        self.L.rth_scene_device_build(self.h, ctx)

    def build(self, split=0):
        self._chk(self.L.rth_build(self.h, split))

    def build_tlas(self, split, instances):
        idx = np.array([i for i, _ in instances], dtype=np.int32)
        T = np.ascontiguousarray(np.stack([np.asarray(t, dtype=np.float32).reshape(16) for _, t in instances]))
        self._chk(self.L.rth_build_tlas(self.h, split, len(idx), _p(idx), _p(T)))

    def set_raytracer(self, rt):
        self.L.rth_set_raytracer(self.h, int(rt))

    def raytracer(self):
        return bool(self.L.rth_get_raytracer(self.h))

    def set_time(self, t):
        """Scene::SetTime (animation + refit on the device); needs a committed scene."""
        self._chk(self.L.rth_scene_set_time(self.h, C.c_float(t)))

    # ---- dumps ----
    def n_meshes(self):
        return int(self.L.rth_meshes(self.h))

    def mesh_tris(self, mesh):
        n = self.L.rth_mesh_count(self.h, mesh)
        if n < 0:
            raise IndexError("mesh %d of %d" % (mesh, self.n_meshes()))
        out = np.zeros((n, 15), dtype=np.float32)
        ids = np.zeros(n, dtype=np.int32)
        self.L.rth_mesh_get(self.h, mesh, _p(out), _p(ids))
        return out, ids

    def bvh_dump_info(self, blas=-1):
        """sizes of a tree without copying it: nodes_used, N (primitives), NTri, NSph, NPla, max_depth"""
        info = (C.c_int * 7)()
        self.L.rth_bvh_info(self.h, blas, info)
        return dict(nodes_used=info[0], N=info[1], NTri=info[2], NSph=info[3], NPla=info[4], max_depth=info[5])

    def bvh_dump(self, blas=-1):
        info = (C.c_int * 7)()
        self.L.rth_bvh_info(self.h, blas, info)
        nodes = np.zeros((info[0], 8), dtype=np.uint32)
        prim = np.zeros(info[1], dtype=np.uint32)
        self.L.rth_bvh_get(self.h, blas, _p(nodes), _p(prim))
        return dict(nodes=nodes, prim_idx=prim, nodes_used=info[0], N=info[1], NTri=info[2], NSph=info[3],
                    NPla=info[4], max_depth=info[5])

    def blas_count(self):
        return self.L.rth_blas_count(self.h)

    def tlas_dump(self):
        n = self.L.rth_tlas_nodes_used(self.h)
        nodes = np.zeros((n, 8), dtype=np.uint32)
        self.L.rth_tlas_get(self.h, _p(nodes))
        return nodes

    def instance_dump(self, i):
        blas = C.c_int()
        T = np.zeros(16, dtype=np.float32)
        iT = np.zeros(16, dtype=np.float32)
        b = np.zeros(6, dtype=np.float32)
        self.L.rth_instance_get(self.h, i, C.byref(blas), _p(T), _p(iT), _p(b))
        return dict(blas=blas.value, T=T, invT=iT, bounds=b)

    def describe(self):
        """Pointer to the rt_scene_desc this scene flattens to (valid until the scene changes)."""
        p = self.L.rth_describe(self.h)
        if not p:
            raise RuntimeError(self.L.rth_last_error(self.h).decode())
        return C.c_void_p(p)

    # single-ray forms with the reference's call shape (one device round trip each)
    def find_nearest_one(self, O, D, tmax=1e34, t_min=1e-6):
        t, obj = C.c_float(), C.c_int()
        n = (C.c_float * 3)()
        self._chk(self.L.rth_scene_find_nearest(self.h, _f3(O), _f3(D), C.c_float(tmax), C.c_float(t_min), C.byref(t), C.byref(obj), n))
        return t.value, obj.value, np.array(list(n), dtype=np.float32)

    def member_intersect(self, which, index, O, D, tmax=1e34):
        """bvh::Intersect (which 0), tlas::Intersect (1), bvhInstance::BIntersect (2) on one ray: (t, objIdx, normal)"""
        t, obj = C.c_float(), C.c_int()
        n = (C.c_float * 3)()
        self._chk(self.L.rth_member_intersect(self.h, which, index, _f3(O), _f3(D), C.c_float(tmax), C.byref(t), C.byref(obj), n))
        return t.value, obj.value, np.array(list(n), dtype=np.float32)

    def member_occluded(self, which, index, O, D, tmax=1e34):
        return bool(self._chk(self.L.rth_member_occluded(self.h, which, index, _f3(O), _f3(D), C.c_float(tmax))))

    def sky_color_one(self, D):
        rgb = (C.c_float * 3)()
        self._chk(self.L.rth_scene_sky_color(self.h, _f3(D), rgb))
        return np.array(list(rgb), dtype=np.float32)

    def is_occluded_one(self, O, D, tmax=1e34):
        return bool(self._chk(self.L.rth_scene_is_occluded(self.h, _f3(O), _f3(D), C.c_float(tmax))))


class HostRenderer:
    """rapt::Renderer (Init / Tick / Trace / Sample) plus direct access to its device context through
    the C ABI of include/rt_amd.h."""

    def __init__(self, width, height, device=0, devices=None):
        """devices: list of HIP devices for rapt::Renderer::UseDevices (one context each; Tick shards the rows over
        them and gathers into context 0).  A device may repeat: [0, 0] runs the multi-context path on one GPU."""
        self.L = host_lib()
        self.rt = rt_lib()
        self.w, self.hgt = width, height
        if devices:
            arr = (C.c_int * len(devices))(*devices)
            self.h = C.c_void_p(self.L.rth_renderer_new_multi(width, height, len(devices), arr))
        else:
            self.h = C.c_void_p(self.L.rth_renderer_new(width, height, device))
        self.scene = HostScene(self.L.rth_renderer_scene(self.h))
        self._chk(self.L.rth_renderer_init(self.h))
        self.ctx = C.c_void_p(self.L.rth_renderer_ctx(self.h))

    def close(self):
        if self.h:
            self.L.rth_renderer_free(self.h)
            self.h = None

    def _chk(self, rc):
        if rc < 0:
            raise RuntimeError(self.L.rth_renderer_error(self.h).decode())
        return rc

    def _rt(self, rc):
        if rc != 0:
            raise RuntimeError("rt_amd error %d: %s" % (rc, self.rt.rt_last_error(self.ctx).decode()))

    def commit(self):
        """Scene::Commit: flatten + rt_upload_scene."""
        self._chk(self.L.rth_renderer_commit(self.h))

    def set_camera(self, cam_pos, top_left, top_right, bottom_left, fisheye=False, view_angle=0.25, y_angle=0.0):
        self.L.rth_renderer_set_camera(self.h, _f3(cam_pos), _f3(top_left), _f3(top_right), _f3(bottom_left),
                                       int(fisheye), C.c_float(view_angle), C.c_float(y_angle))
        self._chk(self.L.rth_renderer_sync_camera(self.h))

    def camera(self):
        out = np.zeros(12, dtype=np.float32)
        self.L.rth_renderer_get_camera(self.h, _p(out))
        return out.reshape(4, 3)

    # ---- Renderer surface ----
    def tick(self):
        self._chk(self.L.rth_renderer_tick(self.h))

    def tick_qlearning(self, grid, lo=(0, 0, 0), hi=(1, 1, 1), alpha=0.3, epsilon=0.2, q_init=1.0):
        """rapt::Renderer::EnableQLearning (grid > 0) / DisableQLearning: Tick then learns between path frames, on every context"""
        self._qgrid = grid
        self._chk(self.L.rth_renderer_qlearning(self.h, grid, _f3(lo), _f3(hi), C.c_float(alpha), C.c_float(epsilon), C.c_float(q_init)))

    def iteration(self):
        """Scene::GetIterationNumber()"""
        return int(self.L.rth_renderer_iteration(self.h))

    def tick_accumulator(self):
        p = C.cast(self.L.rth_renderer_accumulator(self.h), C.POINTER(C.c_float))
        return np.ctypeslib.as_array(p, shape=(self.hgt, self.w, 4)).copy()

    def tick_pixels(self):
        p = C.cast(self.L.rth_renderer_pixels(self.h), C.POINTER(C.c_uint32))
        return np.ctypeslib.as_array(p, shape=(self.hgt, self.w)).copy()

    def trace_one(self, O, D, depth, path=False, energy=(1, 1, 1)):
        rgb = (C.c_float * 3)()
        self._chk(self.L.rth_renderer_trace(self.h, int(path), _f3(O), _f3(D), depth, _f3(energy), rgb))
        return np.array(list(rgb), dtype=np.float32)

    # ---- C ABI, direct ----
    def render(self, mode, frame0=0, nframes=1, seed_base=0x12345678, y0=0, y1=None, max_depth=4):
        self._rt(self.rt.rt_render(self.ctx, mode, frame0, nframes, seed_base, y0, self.hgt if y1 is None else y1, max_depth))

    def render_rows(self, mode, frame0, nframes, row_first, row_stride, row_count, seed_base=0x12345678, max_depth=4):
        self._rt(self.rt.rt_render_rows(self.ctx, mode, frame0, nframes, seed_base, row_first, row_stride, row_count, max_depth))

    def set_time(self, t):
        """Scene::SetTime with animation on: deform + refit on the GPU."""
        self._rt(self.rt.rt_set_time(self.ctx, t))

    def clear(self):
        self._rt(self.rt.rt_clear(self.ctx))

    def synchronize(self):
        self._rt(self.rt.rt_synchronize(self.ctx))

    def accumulator(self, y0=0, y1=None):
        y1 = self.hgt if y1 is None else y1
        out = np.zeros((y1 - y0, self.w, 4), dtype=np.float32)
        self._rt(self.rt.rt_download_accumulator(self.ctx, y0, y1, _p(out)))
        return out

    def resolve(self, it=1, y0=0, y1=None):
        y1 = self.hgt if y1 is None else y1
        out = np.zeros((y1 - y0, self.w), dtype=np.uint32)
        self._rt(self.rt.rt_resolve(self.ctx, it, y0, y1, _p(out)))
        return out

    def accumulator_device_ptr(self):
        return self.rt.rt_accumulator_device_ptr(self.ctx)

    def bind_accumulator(self, device_ptr):
        self._rt(self.rt.rt_bind_accumulator(self.ctx, C.c_void_p(device_ptr)))

    def find_nearest(self, O, D, tmax=None, t_min=1e-6):
        O = np.ascontiguousarray(O, dtype=np.float32).reshape(-1, 3)
        D = np.ascontiguousarray(D, dtype=np.float32).reshape(-1, 3)
        out = np.zeros(len(O), dtype=RT_HIT_DTYPE)
        tm = None if tmax is None else _p(np.ascontiguousarray(tmax, dtype=np.float32))
        self._rt(self.rt.rt_intersect_batch(self.ctx, len(O), _p(O), _p(D), tm, t_min, _p(out)))
        return dict(t=out["t"].copy(), obj=out["obj_idx"].copy(), mat=out["material"].copy(), normal=out["normal"].copy())

    def is_occluded(self, O, D, tmax=None):
        O = np.ascontiguousarray(O, dtype=np.float32).reshape(-1, 3)
        D = np.ascontiguousarray(D, dtype=np.float32).reshape(-1, 3)
        out = np.zeros(len(O), dtype=np.uint8)
        tm = None if tmax is None else _p(np.ascontiguousarray(tmax, dtype=np.float32))
        self._rt(self.rt.rt_occluded_batch(self.ctx, len(O), _p(O), _p(D), tm, _p(out)))
        return out

    def build_tlas(self, bounds6):
        """rt_build_tlas: tlas::build on the device; bounds6 (n, 6) instance world boxes.  Returns the nodes as (k, 8) uint32."""
        b = np.ascontiguousarray(bounds6, dtype=np.float32).reshape(-1, 6)
        nodes = np.zeros((2 * len(b) + 1, 8), dtype=np.uint32)
        used = C.c_uint32(0)
        self._rt(self.rt.rt_build_tlas(self.ctx, _p(b), len(b), _p(nodes), C.byref(used)))
        return nodes[:used.value].copy()

    def build_bvh(self, tri_v9=None, spheres=None, planes=None, split=0):
        """rt_build_bvh_split: the reference's bvh::Build on the device (split: bvh.h:38-43, 0 = binned SAH).  tri_v9: (n, 9) vertices,
        spheres: (n, 4) pos + r, planes: (n, 4) N + d.  Returns (nodes[:nodes_used] as an (n, 8) uint32 view
        like the oracle's dump, prim_idx)."""
        tv = np.zeros((0, 9), np.float32) if tri_v9 is None else np.ascontiguousarray(tri_v9, dtype=np.float32).reshape(-1, 9)
        sp = np.zeros((0, 4), np.float32) if spheres is None else np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 4)
        pl = np.zeros((0, 4), np.float32) if planes is None else np.ascontiguousarray(planes, dtype=np.float32).reshape(-1, 4)
        T = np.zeros(len(tv), dtype=RT_TRIANGLE_DTYPE)
        T["v0"], T["v1"], T["v2"] = tv[:, 0:3], tv[:, 3:6], tv[:, 6:9]
        S = np.zeros(len(sp), dtype=RT_SPHERE_DTYPE)
        S["pos"], S["r"] = sp[:, :3], sp[:, 3]
        P = np.zeros(len(pl), dtype=RT_PLANE_DTYPE)
        P["N"], P["d"] = pl[:, :3], pl[:, 3]
        n = len(T) + len(S) + len(P)
        nodes = np.zeros(2 * (n + 1), dtype=RT_BVH_NODE_DTYPE)
        prim = np.zeros(max(n, 1), dtype=np.uint32)
        used = C.c_uint32(0)
        self._rt(self.rt.rt_build_bvh_split(self.ctx, int(split), _p(T) if len(T) else None, len(T), _p(S) if len(S) else None, len(S),
                                      _p(P) if len(P) else None, len(P), _p(nodes), _p(prim), C.byref(used)))
        return nodes[:used.value].view(np.uint32).reshape(-1, 8).copy(), prim[:n].copy()

    def primary_hits(self, t_min=1e-6):
        obj = np.zeros((self.hgt, self.w), dtype=np.int32)
        t = np.zeros((self.hgt, self.w), dtype=np.float32)
        self._rt(self.rt.rt_primary_hits(self.ctx, t_min, _p(obj), _p(t)))
        return obj, t

    def set_scene_raytracer(self, flag):
        """rt_set_scene_raytracer: -1 the flag follows the function (Trace: set, Sample: clear), 0 / 1 scene.raytracer as the caller holds it"""
        self._rt(self.rt.rt_set_scene_raytracer(self.ctx, int(flag)))

    def trace_batch(self, mode, O, D, depth=4, seed_base=0x12345678, energy=(1, 1, 1)):
        O = np.ascontiguousarray(O, dtype=np.float32).reshape(-1, 3)
        D = np.ascontiguousarray(D, dtype=np.float32).reshape(-1, 3)
        out = np.zeros((len(O), 3), dtype=np.float32)
        self._rt(self.rt.rt_trace_batch_energy(self.ctx, mode, len(O), _p(O), _p(D), depth, seed_base, _f3(energy), _p(out)))
        return out

    def scope_nearest(self, scope, index, O, D, tmax=None):
        """rt_intersect_scope: 1 the accelerator alone, 2 BLAS index, 3 instance index"""
        O = np.ascontiguousarray(O, dtype=np.float32).reshape(-1, 3)
        D = np.ascontiguousarray(D, dtype=np.float32).reshape(-1, 3)
        out = np.zeros(len(O), dtype=RT_HIT_DTYPE)
        tm = None if tmax is None else _p(np.ascontiguousarray(tmax, dtype=np.float32))
        self._rt(self.rt.rt_intersect_scope(self.ctx, scope, index, len(O), _p(O), _p(D), tm, C.c_float(0.0), _p(out)))
        return dict(t=out["t"].copy(), obj=out["obj_idx"].copy(), mat=out["material"].copy(), normal=out["normal"].copy())

    def scope_occluded(self, scope, index, O, D, tmax=None):
        O = np.ascontiguousarray(O, dtype=np.float32).reshape(-1, 3)
        D = np.ascontiguousarray(D, dtype=np.float32).reshape(-1, 3)
        out = np.zeros(len(O), dtype=np.uint8)
        tm = None if tmax is None else _p(np.ascontiguousarray(tmax, dtype=np.float32))
        self._rt(self.rt.rt_occluded_scope(self.ctx, scope, index, len(O), _p(O), _p(D), tm, _p(out)))
        return out

    def sky_color(self, D):
        D = np.ascontiguousarray(D, dtype=np.float32).reshape(-1, 3)
        out = np.zeros((len(D), 3), dtype=np.float32)
        self._rt(self.rt.rt_sky_color_batch(self.ctx, len(D), _p(D), _p(out)))
        return out

    def set_counting(self, on):
        """False / 0 off, True / RT_COUNT_REFERENCE the reference's walk, RT_COUNT_EXECUTED the timed kernels' walk"""
        self._rt(self.rt.rt_set_counting(self.ctx, int(on)))

    def counters(self, reset=True):
        c = np.zeros(8, dtype=np.uint64)
        self._rt(self.rt.rt_get_counters(self.ctx, _p(c), int(reset)))
        return dict(zip(COUNTER_NAMES, [int(x) for x in c]))

    def counters_split(self, reset=True):
        a = np.zeros(8, dtype=np.uint64)
        b = np.zeros(8, dtype=np.uint64)
        self._rt(self.rt.rt_get_counters_split(self.ctx, _p(a), _p(b), int(reset)))
        return dict(zip(COUNTER_NAMES, [int(x) for x in a])), dict(zip(COUNTER_NAMES, [int(x) for x in b]))

    # ---- Q-learning guided sampler (include/rt_amd.h rt_qlearn_*; no reference code: parity unpinned) ----
    def qlearn_enable(self, grid, lo, hi, alpha=0.3, epsilon=0.2, q_init=1.0, learn_mask=0):
        p = RtQlearnParams(grid, (C.c_float * 3)(*lo), (C.c_float * 3)(*hi), alpha, epsilon, q_init, learn_mask)
        self._qgrid = grid
        self._rt(self.rt.rt_qlearn_enable(self.ctx, C.byref(p)))

    def qlearn_disable(self):
        self._rt(self.rt.rt_qlearn_enable(self.ctx, None))

    def qlearn_apply(self):
        self._rt(self.rt.rt_qlearn_apply(self.ctx))

    def qlearn_sums(self):
        n = self._qgrid ** 3
        sums, cnts = np.zeros((n, 64), np.int64), np.zeros((n, 64), np.uint32)
        self._rt(self.rt.rt_qlearn_get_sums(self.ctx, _p(sums), _p(cnts)))
        return sums, cnts

    def qlearn_set_sums(self, sums, cnts):
        sums, cnts = np.ascontiguousarray(sums, np.int64), np.ascontiguousarray(cnts, np.uint32)
        self._rt(self.rt.rt_qlearn_set_sums(self.ctx, _p(sums), _p(cnts)))

    def qlearn_bind_sums(self, sums_ptr, counts_ptr):
        """rt_qlearn_bind_sums: the pending reward sums live in the caller's device arrays (data_ptr() of an int64 / int32 torch tensor
        of grid^3 * 64 elements) from here on"""
        self._rt(self.rt.rt_qlearn_bind_sums(self.ctx, C.c_void_p(sums_ptr), C.c_void_p(counts_ptr)))

    def qlearn_table(self):
        tab = np.zeros((self._qgrid ** 3, 64), np.float32)
        self._rt(self.rt.rt_qlearn_get_table(self.ctx, _p(tab)))
        return tab

    def build_info(self):
        """rt_build_info() + rt_tuning_info(): compile flags / compile-time tuning of the library and the tuning this context resolved"""
        return self.rt.rt_build_info().decode() + " | " + self.rt.rt_tuning_info(self.ctx).decode()

    def set_profiling(self, on):
        self._rt(self.rt.rt_set_profiling(self.ctx, int(on)))

    def profile(self, reset=True):
        p = RtProfile()
        self._rt(self.rt.rt_get_profile(self.ctx, C.byref(p), int(reset)))
        return {k: dict(launches=int(getattr(p, k).launches), ms=float(getattr(p, k).ms)) for k in ("generate", "extend", "shade", "connect", "query")}


def device_pci_bus_id(device):
    """rt_device_pci_bus_id: the PCI address of HIP device 'device' (what two ranks on one GPU have in common)"""
    buf = C.create_string_buffer(64)
    rc = rt_lib().rt_device_pci_bus_id(int(device), buf, 64)
    if rc != 0:
        raise RuntimeError("rt_amd error %d: %s" % (rc, rt_lib().rt_last_error(None).decode()))
    return buf.value.decode()


def algorithmic_bytes(counters, executed=False):
    """Algorithmic bytes of a set of traversals, SURVEY.md section 8(d) literally: 64 B per BLAS inner-node visit
    (two 32-B children), 52 B per primitive test made BY THE TRAVERSAL (4-B index + 48-B triangle), 48 B per ray
    (32 in, 16 out), 64 B per TLAS inner visit and 128 B per instance entry (two mat4).  The head tests of
    Scene::FindNearest (lights, brute-force spheres / planes) run where rays are created, not in the traversal
    kernels, and are not priced here; neither are the 48-B reach records the timed kernels read per TLAS visit
    (bench.py reports those apart).  'executed' is accepted for older callers and ignored."""
    rays = counters["rays_nearest"] + counters["rays_occluded"]
    return (64 * counters["inner_visits"] + 52 * counters["prim_tests"] + 48 * rays
            + 64 * counters["tlas_inner"] + 128 * counters["instance_visits"])
