"""Scene definitions: the reference's scene factories (template/scene.h:791-1209) expressed as data,
made concrete for the five BASELINE.json configurations as SURVEY.md section 8(d) prescribes.

Every function takes a *builder* `b` (the product's host_api.HostScene or, in tests, the oracle's
OracleScene -- both implement the same protocol) and returns a dict describing how to render it:
resolution, integrator, number of frames, camera override.  Constructor argument orders follow
the reference classes (diffuse: template/scene.h:595; metal :629; glass :643; AreaLight :97;
Sphere :350; Plane :404; Mesh :261, :285).
"""
import math
import numpy as np
from . import assets

BINNEDSAH, SAMESIZE, LONGESTAXIS, SAH = 0, 1, 2, 3
PI_F = float(np.float32(3.14159265358979323846264))


def _c255(r, g, b):
    """float3(r, g, b) / 255 in float32 (template/scene.h:1390-1396)."""
    return tuple(float(np.float32(x) / np.float32(255)) for x in (r, g, b))


WHITE = (1.0, 1.0, 1.0)
RED = _c255(255, 0, 0)
BLUE = _c255(0, 0, 255)
BABYBLUE = (float(np.float32(0.6)), float(np.float32(0.6)), 1.0)
GREEN = _c255(0, 255, 0)
GOLD = _c255(218, 165, 32)
PINK = _c255(255, 20, 147)


def _f32(x):
    return float(np.float32(x))


def background_scene(b, rt=True, sky=True, split=BINNEDSAH):
    """instantiateBackgroundScene (template/scene.h:791-813): the reference's default scene.
    hdr.hdr is missing from the snapshot, so a synthetic sky stands in."""
    if sky:
        b.sky(assets.synthetic_sky())
    b.area_light(11, (4.5, 5.0, 7.0), 19.0, WHITE, 2.0, (0, -1, 0))
    orange = b.glass(1.5, _c255(212, 34, 93), rt=rt)
    red_glass = b.glass(1.5, RED, rt=rt)
    b.glass(1.5, _c255(105, 5, 255), rt=rt)  # pinkGlass (unused by any primitive)
    green_glass = b.glass(1.5, _c255(12, 207, 135), rt=rt)
    blue_glass = b.glass(1.5, _c255(30, 30, 232), rt=rt)
    floor = b.diffuse(0.8, WHITE, 0.0, 1.0, 4, rt=rt)
    b.mesh_obj(10, assets.obj_path("ico"), orange, (4.5, 0.5, 0.0), 0.5)
    b.sphere(1, red_glass, (3.0, 0.5, 0), 0.5)
    b.mesh_obj(4, assets.obj_path("ico"), blue_glass, (1.5, 0.5, 0.0), 0.5)
    b.sphere(3, green_glass, (0.0, 0.5, 0), 0.5)
    b.plane(0, floor, (0, 1, 0), 0)
    b.build(split)
    return dict(name="background", tlas=False)


def config1(b):
    """BASELINE config 1: ico.obj scene, 256x256, 1 frame, Whitted, primary rays only."""
    d = background_scene(b)
    d.update(width=256, height=256, mode="whitted", frames=1, primary_only=True)
    return d


def scene3(b, force_diffuse=True, split=SAH, rt=True):
    """instantiateScene3 (template/scene.h:1045-1092) without the cubes (never intersected:
    template/scene.h:1253-1255, bvh.cpp:20-22).  force_diffuse=True gives BASELINE config 2
    ('diffuse + area light'): every material diffuse and only the area light."""
    b.sky(assets.synthetic_sky(seed=11))
    if force_diffuse:
        def glass(ir, col):
            return b.diffuse(0.8, col, 0.6, 0.4, 2, rt=rt)
        def metal(f, col):
            return b.diffuse(0.8, col, 0.6, 0.4, 2, rt=rt)
    else:
        def glass(ir, col):
            return b.glass(ir, col, rt=rt)
        def metal(f, col):
            return b.metal(f, col, rt=rt)
    standard_glass = glass(1.5, WHITE)
    pink_glass = glass(1.5, PINK)
    green_diff = b.diffuse(0.8, GREEN, 0.6, 0.4, 2, rt=rt)
    green_metal = metal(0.7, GREEN)
    gold_metal = metal(0.7, GOLD)
    blue_metal = metal(0.7, BLUE)
    b.area_light(11, (0.1, 3, 1.5), 10.0, WHITE, 1.0, (0, -1, 0))
    if not force_diffuse:
        b.dir_light(12, (5, 3, -1), 10.0, WHITE, (-1, -1, 1), 1)
    floor = b.diffuse(0.8, RED, 0.0, 1.0, 4, rt=rt)
    b.plane(0, floor, (0, 1, 0), 1)
    three_pos = np.array([0, 0, 2], dtype=np.float32)
    three_scale = np.float32(2.5)
    b.mesh_obj(1, assets.obj_path("three"), green_diff, (0, 0, 2), 2.5)
    sph = [
        (1, blue_metal, (0.410241, -0.085121, -0.122131)), (2, pink_glass, (0.122131, -0.085121, 0.410241)),
        (3, blue_metal, (-0.410241, -0.085121, 0.122131)), (4, standard_glass, (-0.122131, -0.085121, -0.410241)),
        (5, green_metal, (0.500000, -0.367977, -0.001909)), (6, pink_glass, (0.001909, -0.367977, 0.500000)),
        (7, green_metal, (-0.500000, -0.367977, 0.001909)), (8, standard_glass, (-0.001909, -0.367977, -0.500000)),
        (8, blue_metal, (0.236091, 0.198982, -0.236091)), (8, pink_glass, (0.236091, 0.198982, 0.236091)),
        (8, green_metal, (-0.236091, 0.198982, 0.236091)), (8, standard_glass, (-0.236091, 0.198982, -0.236091)),
    ]
    for idx, mat, p in sph:
        # float3(p) * threeScale + threePos - float3(0, 0.05f, 0), in float32 (template/scene.h:1073)
        q = np.array(p, dtype=np.float32) * three_scale + three_pos - np.array([0, 0.05, 0], dtype=np.float32)
        b.sphere(idx, mat, tuple(float(x) for x in q), 0.05)
    q = np.array((0.0, 0.561019, 0.0), dtype=np.float32) * three_scale + three_pos + np.array([0, 0.25, 0], dtype=np.float32)
    b.mesh_obj(2, assets.obj_path("stellatedDode"), gold_metal, tuple(float(x) for x in q), 0.4)
    b.build(split)
    return dict(name="scene3", tlas=False)


def config2(b):
    """BASELINE config 2: three.obj scene, full-sweep SAH, 1920x1080, 16 frames, path mode."""
    d = scene3(b, force_diffuse=True, split=SAH)
    d.update(width=1920, height=1080, mode="path", frames=16)
    return d


def pretty_tlas(b, n_instances=8, rt=True, split=BINNEDSAH):
    """BASELINE config 3: instantiatePrettyScene1's contents (template/scene.h:815-840) with the
    unity.tri mesh held as BLASes and instanced through the TLAS.  Two meshes share the triangle
    data: one glass (dielectric), one metal (mirror); instances alternate between them.
    n_instances = 8 gives 8 x 12,584 = 100,672 instanced triangles ('~100k tris')."""
    b.sky(assets.synthetic_sky(seed=5))
    b.area_light(11, (0.1, 4.0, 5.0), 8.0, WHITE, 1.0, (0, -1, 0))
    b.area_light(12, (0.1, 4.0, 3.0), 10.0, WHITE, 1.0, (0, -1, 0))
    light_diff = b.diffuse(0.8, WHITE, 0.6, 0.4, 1200, emission=1.2, rt=rt)
    standard_glass = b.glass(1.5, WHITE, rt=rt)
    blue_diff = b.diffuse(0.8, BLUE, 0.6, 0.4, 10, rt=rt)
    blue_glass = b.glass(1.5, BABYBLUE, rt=rt)
    yellow_metal = b.metal(0.7, GOLD, rt=rt)
    pink_metal = b.metal(0.7, PINK, rt=rt)
    standard_metal = b.metal(0.7, WHITE, rt=rt)
    b.sphere(7, blue_diff, (-0.7, -0.5, 2.0), 0.5)
    b.sphere(8, blue_glass, (-1.9, -0.5, 2.0), 0.5)
    b.sphere(9, yellow_metal, (-3.1, -0.5, 2.0), 0.5)
    b.sphere(6, pink_metal, (-4.3, -0.5, 2.0), 0.5)
    b.plane(1, light_diff, (0, 1, 0), 1)
    m_glass = b.mesh_tri(2, assets.tri_path("unity"), standard_glass)
    m_metal = b.mesh_tri(3, assets.tri_path("unity"), standard_metal)
    inst = []
    for i in range(n_instances):
        col, row = i % 4, i // 4
        t = (_f32(-3.0 + 2.6 * col), _f32(0.25), _f32(4.0 + 3.0 * row))
        ry = _f32(PI_F * 0.25 * i)
        T = b.trs(t, _f32(0.8), 0.0, ry, 0.0)
        inst.append((m_glass if i % 2 == 0 else m_metal, T))
    b.build_tlas(split, inst)
    cam = dict(cam_pos=(0.0, 2.0, -3.0), top_left=(-16 / 9, 3.0, -1.0), top_right=(16 / 9, 3.0, -1.0), bottom_left=(-16 / 9, 1.0, -1.0))
    return dict(name="pretty_tlas", tlas=True, camera=cam)


def config3(b):
    d = pretty_tlas(b, 8)
    d.update(width=1920, height=1080, mode="path", frames=64)
    return d


TOWER_TRIANGLES = 51200  # assets.lattice_tower(): 160 levels x 40 sides x 8 triangles


def tower_scene(b, rt=True, levels=160, sides=40):
    """BASELINE config 4: instantiateEifelScene (template/scene.h:870-878) with a procedural
    lattice tower (51,200 triangles by default) standing in for the missing eifel.obj, and a synthetic sky."""
    b.sky(assets.synthetic_sky(seed=21))
    light_diff = b.diffuse(0.8, WHITE, 0.6, 0.4, 1200, emission=1.2, rt=rt)
    b.plane(2, light_diff, (0, 1, 0), 1)
    b.area_light(11, (0.1, 7.0, 5.0), 8.0, WHITE, 1.0, (0, -1, 0))
    standard_metal = b.metal(0.7, WHITE, rt=rt)
    tris = assets.lattice_tower(levels, sides).reshape(-1, 3)
    tris = (tris * np.float32(0.8) + np.array([0, -1.0, 5.0], dtype=np.float32)).astype(np.float32).reshape(-1, 9)
    b.mesh_raw(2, standard_metal, tris)
    b.build(BINNEDSAH)
    return dict(name="tower", tlas=False)


def config4(b):
    """BASELINE config 4: the 51,200-triangle tower stand-in + synthetic sky, 1920x1080, 256 frames, path mode."""
    d = tower_scene(b)
    d.update(width=1920, height=1080, mode="path", frames=256, triangles=TOWER_TRIANGLES)
    return d


def bigb_instanced(b, n=16, rt=True, mesh="BigB"):
    """BASELINE config 5: BigB.obj (11,830 triangles) instanced n times on a grid through
    bvhInstance/TLAS, transforms Translate*Scale*RotateY as in TLASSceneTest2
    (template/scene.h:941-972).  The Q-learning sampler named by BASELINE.json has no code in the
    reference snapshot and is not part of this scene."""
    b.sky(assets.synthetic_sky(seed=9))
    red_diff = b.diffuse(0.8, RED, 0.0, 1, 1, rt=rt)
    gold_diff = b.diffuse(0.8, GOLD, 0.8, 0.2, 1, rt=rt)
    floor = b.diffuse(0.8, WHITE, 0.0, 1.0, 4, rt=rt)
    b.area_light(11, (0, 6.0, 0), 16.0, WHITE, 2.0, (0, -1, 0))
    m = b.mesh_obj(1, assets.obj_path(mesh), red_diff, (0, 0.5, 0), 1)
    side = int(math.ceil(math.sqrt(n)))
    inst = []
    for i in range(n):
        cx, cz = i % side, i // side
        t = (_f32(-4.5 + 3.0 * cx), 0.0, _f32(3.0 + 3.0 * cz))
        T = b.trs(t, 4.0, 0.0, _f32(PI_F * 0.5 * (i % 4)), 0.0)
        inst.append((m, T))
    b.plane(0, floor, (0, 1, 0), 0)
    b.sphere(7, gold_diff, (1.8, -0.5, 2.0), 0.5)
    b.build_tlas(BINNEDSAH, inst)
    cam = dict(cam_pos=(0.0, 5.0, -6.0), top_left=(-16 / 9, 5.6, -4.2), top_right=(16 / 9, 5.6, -4.2), bottom_left=(-16 / 9, 3.8, -5.1))
    return dict(name="bigb_instanced", tlas=True, camera=cam)


def config5(b):
    d = bigb_instanced(b, 16)
    d.update(width=3840, height=2160, mode="path", frames=1024)
    return d


def tlas_test2(b, rt=True, mesh="lowBigB"):
    """TLASSceneTest2 (template/scene.h:941-972): one mesh, three instances, floor plane and one
    sphere tested by brute force.  `mesh` = 'BigB' is the reference's choice; 'lowBigB' keeps
    CPU tests fast."""
    b.sky(assets.synthetic_sky(seed=2))
    red_diff = b.diffuse(0.8, RED, 0.0, 1, 1, rt=rt)
    gold_diff = b.diffuse(0.8, GOLD, 0.8, 0.2, 1, rt=rt)
    floor = b.diffuse(0.8, WHITE, 0.0, 1.0, 4, rt=rt)
    b.area_light(11, (0, 6.0, 0), 16.0, WHITE, 2.0, (0, -1, 0))
    m = b.mesh_obj(1, assets.obj_path(mesh), red_diff, (0, 0.5, 0), 1)
    T0 = b.trs((0, 0, 3), 4, 0.0, _f32(PI_F * 0.5), 0.0)
    T1 = b.trs((2, 0, 3), 4, 0.0, PI_F, 0.0)
    T2 = b.trs((_f32(-2.3), 0, 3), 4, 0.0, _f32(PI_F * 0.5), 0.0)
    b.plane(0, floor, (0, 1, 0), 0)
    b.sphere(7, gold_diff, (1.8, -0.5, 2.0), 0.5)
    b.build_tlas(BINNEDSAH, [(m, T0), (m, T1), (m, T2)])
    return dict(name="tlas_test2", tlas=True)


def mixed_small(b, rt=True, split=BINNEDSAH):
    """Small all-materials scene for fast parity tests: glass + metal + diffuse meshes and
    spheres, two area lights, floor; exercises every shading branch."""
    b.sky(assets.synthetic_sky(64, 32, seed=4))
    b.area_light(11, (1.0, 4.0, 1.0), 10.0, WHITE, 1.0, (0, -1, 0))
    b.area_light(12, (-1.0, 3.0, 0.5), 5.0, WHITE, 0.5, (0, -1, 0))
    gl = b.glass(1.5, BABYBLUE, (0.1, 0.2, 0.05), rt=rt)
    me = b.metal(0.7, GOLD, rt=rt)
    df = b.diffuse(0.8, GREEN, 0.6, 0.4, 10, rt=rt)
    fl = b.diffuse(0.8, WHITE, 0.0, 1.0, 4, rt=rt)
    b.mesh_obj(1, assets.obj_path("ico"), gl, (-0.9, 0.6, 0.6), 0.5)
    b.mesh_obj(2, assets.obj_path("stellatedDode"), me, (0.9, 0.7, 0.8), 0.5)
    b.mesh_obj(3, assets.obj_path("three"), df, (0.0, 0.5, 1.8), 1.2)
    b.sphere(1, gl, (0.2, 0.35, 0.2), 0.35)
    b.sphere(2, me, (-0.5, 0.25, -0.3), 0.25)
    b.plane(0, fl, (0, 1, 0), 0)
    b.build(split)
    return dict(name="mixed_small", tlas=False)


REGISTRY = {
    "config1": config1, "config2": config2, "config3": config3, "config4": config4, "config5": config5,
    "background": background_scene, "scene3": scene3, "pretty_tlas": pretty_tlas, "tower": tower_scene,
    "bigb_instanced": bigb_instanced, "tlas_test2": tlas_test2, "mixed_small": mixed_small,
}
