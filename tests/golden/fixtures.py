"""Definitions shared by the golden-vector generator (make_golden.py, which runs the oracle) and the tests that
read the vectors (which must not need the oracle): fixture scene list, key naming, the ray generator and
the single-primitive scenes.  Nothing here imports the oracle or the product."""
import numpy as np

RNG_SEEDS = (0, 1, 2, 61, 0x12345678, 0xDEADBEEF, 0xFFFFFFFE, 1768515948)  # the last one hashes to 0 (replaced state)
FIXTURE_SCENES = (
    ("background", {}), ("mixed_small", {}), ("scene3", {"force_diffuse": False, "split": 0}),
    ("scene3", {"force_diffuse": True, "split": 3}), ("tlas_test2", {}), ("bigb_instanced", {"n": 16, "mesh": "lowBigB"}),
)


def scene_key(name, kw):
    return name + "".join("_%s%s" % (k, v) for k, v in sorted(kw.items()))


def unit_rays(rng, n, center, spread):
    O = (rng.uniform(-1, 1, (n, 3)) * spread + np.asarray(center)).astype(np.float32)
    tgt = (rng.uniform(-1, 1, (n, 3)) * 0.8).astype(np.float32)
    D = tgt - O
    D[: n // 8] = rng.normal(size=(n // 8, 3))  # some that mostly miss
    D[n // 8: n // 8 + 6] = np.eye(3, dtype=np.float32).repeat(2, 0) * np.array([1, -1] * 3, np.float32)[:, None]  # axis aligned
    D = (D / np.linalg.norm(D.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
    return O, D


def single_primitive(b, kind):
    """One primitive of each type the leaf dispatch knows (bvh.cpp:616-629) + the disk light, as a scene of its own."""
    m = b.diffuse(0.8, (1, 1, 1))
    if kind == "triangle":  # Triangle::Intersect / IsOccluding (template/scene.h:190-237)
        b.mesh_raw(1, m, np.array([[-0.9, -0.7, 0.1, 0.8, -0.6, -0.2, 0.05, 0.9, 0.3]], np.float32))
    elif kind == "sphere":  # Sphere::Intersect / IsOccluding (:351-381)
        b.sphere(5, m, (0.1, -0.05, 0.2), 0.75)
    elif kind == "plane":  # Plane::Intersect (:405-415), a normal the plane AABB quirk treats as unbounded (Q1)
        b.plane(3, m, (0.0, 0.6, 0.8), 0.25)
    elif kind == "disk":  # AreaLight::Intersect (:105-120), Q6
        b.mesh_raw(1, m, np.array([[50, 50, 50, 51, 50, 50, 50, 51, 50]], np.float32))  # far away: only the light is hit
        b.area_light(11, (0.0, 0.4, 0.0), 7.0, (1, 1, 1), 0.8, (0, -1, 0))
    b.build(0)
