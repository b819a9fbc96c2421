"""Writer for 'rapt-scene 1' scene description files (read by rapt::Scene::LoadFile,
host/scene_file.cpp).  SceneWriter implements the scene-builder protocol of scenes.py, so any scene
definition can be recorded to a file:  w = SceneWriter(path); scenes.config3(w); w.close()."""
import os
import numpy as np


def _f(x):
    return "%.9g" % float(np.float32(x))


def _v(v):
    return " ".join(_f(x) for x in v)


class SceneWriter:
    def __init__(self, path):
        self.path = path
        self.dir = os.path.dirname(os.path.abspath(path))
        self.lines = ["rapt-scene 1"]
        self.n_mat = self.n_mesh = self.n_blob = 0
        self.closed = False

    def _blob(self, data, ext):
        name = "%s.%d.%s" % (os.path.splitext(os.path.basename(self.path))[0], self.n_blob, ext)
        self.n_blob += 1
        with open(os.path.join(self.dir, name), "wb") as f:
            f.write(data)
        return name

    def _mat(self, line):
        self.lines.append(line)
        self.n_mat += 1
        return self.n_mat - 1

    def diffuse(self, albedo, col, ks=0.2, kd=0.8, n=2, emission=0.0, shininess=0.0, rt=True):
        a = albedo if hasattr(albedo, "__len__") else (albedo,) * 3
        return self._mat("material diffuse %s %s %s %s %d %s %s %d" % (_v(a), _v(col), _f(ks), _f(kd), n, _f(emission), _f(shininess), int(rt)))

    def metal(self, fuzzy, col, rt=True):
        return self._mat("material metal %s %s %d" % (_f(fuzzy), _v(col), int(rt)))

    def glass(self, ir, col, absorption=(0, 0, 0), rt=True):
        return self._mat("material glass %s %s %s %d" % (_f(ir), _v(col), _v(absorption), int(rt)))

    def area_light(self, idx, pos, strength, col, radius, normal):
        self.lines.append("light area %d %s %s %s %s %s" % (idx, _v(pos), _f(strength), _v(col), _f(radius), _v(normal)))

    def dir_light(self, idx, pos, strength, col, normal, r):
        self.lines.append("light dir %d %s %s %s %s %s" % (idx, _v(pos), _f(strength), _v(col), _v(normal), _f(r)))

    def sphere(self, idx, mat, pos, r):
        self.lines.append("sphere %d %d %s %s" % (idx, mat, _v(pos), _f(r)))

    def plane(self, idx, mat, N, d):
        self.lines.append("plane %d %d %s %s" % (idx, mat, _v(N), _f(d)))

    def _mesh(self, line):
        self.lines.append(line)
        self.n_mesh += 1
        return self.n_mesh - 1

    def mesh_raw(self, group, mat, v9):
        v9 = np.ascontiguousarray(v9, dtype=np.float32).reshape(-1, 9)
        return self._mesh("mesh raw %d %d %s" % (group, mat, self._blob(v9.tobytes(), "tris")))

    def mesh_obj(self, group, path, mat, pos, scale):
        return self._mesh("mesh obj %d %d %s %s %s" % (group, mat, _v(pos), _f(scale), path))

    def mesh_tri(self, group, path, mat):
        return self._mesh("mesh tri %d %d %s" % (group, mat, path))

    def sky(self, pixels):
        px = np.ascontiguousarray(pixels, dtype=np.uint8)
        h, w, n = px.shape
        self.lines.append("sky raw %s %d %d %d" % (self._blob(px.tobytes(), "sky"), w, h, n))

    def trs(self, t, s, rx, ry, rz):
        # the transform product is host arithmetic (mat4 products in float32): take it from the host library
        from . import host_api
        hs = host_api.HostScene()
        T = hs.trs(t, s, rx, ry, rz)
        hs.close()
        return T

    def build(self, split=0):
        self.lines.append("build bvh %d" % split)
        self.close()

    def build_tlas(self, split, instances):
        for mesh, T in instances:
            self.lines.append("instance %d %s" % (mesh, _v(np.asarray(T, dtype=np.float32).reshape(16))))
        self.lines.append("build tlas %d" % split)
        self.close()

    def close(self):
        if not self.closed:
            with open(self.path, "w") as f:
                f.write("\n".join(self.lines) + "\n")
            self.closed = True
