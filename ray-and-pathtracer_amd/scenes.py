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


def terrain_scene(b, n=2048, rt=True):
    """The out-of-cache profile scene (profiles/out_of_cache.py; not a BASELINE configuration): assets.terrain(n) -- 2 n^2 triangles,
    8.4 M at the default -- as one diffuse mesh under the synthetic sky and one area light, one scene BVH (binned SAH), the camera
    above a corner looking across the whole field."""
    b.sky(assets.synthetic_sky(seed=3))
    ground = b.diffuse(0.8, WHITE, 0.2, 0.8, 4, rt=rt)
    b.area_light(11, (0.0, 40.0, 0.0), 600.0, WHITE, 4.0, (0, -1, 0))
    b.mesh_raw(1, ground, assets.terrain(n))
    b.build(BINNEDSAH)
    cam = dict(cam_pos=(-30.0, 14.0, -30.0), top_left=(-29.7073, 14.4434, -27.1932), top_right=(-27.1932, 14.4434, -29.7073), bottom_left=(-30.0764, 12.5127, -27.5623))  # looks at (4, 1, 4), 16:9, screen at distance 2
    return dict(name="terrain", tlas=False, camera=cam, triangles=2 * n * n)


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


def qlearn_probe(b, rt=True):
    """mixed_small lit by the sky and two DirectionalLights instead of its area lights.  No ray can hit a DirectionalLight
    (AreaLight::Intersect is the only light test, template/scene.h:105-120), so no path value is ever +inf (Q7) and the
    linear-radiance mean of Sample() is well defined: the scene of the Q-learning sampler's unbiasedness test."""
    b.sky(assets.synthetic_sky(64, 32, seed=4))
    b.dir_light(11, (5, 3, -1), 6.0, WHITE, (-1, -1, 1), 1)
    b.dir_light(12, (-4, 5, 2), 3.0, WHITE, (1, -1, -0.5), 1)
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
    b.build(BINNEDSAH)
    return dict(name="qlearn_probe", tlas=False)


# ---- the remaining factories of template/scene.h:791-1209, as data -----------------------------------
# Missing assets (.MISSING_LARGE_BLOBS: eifel.obj, christ.obj, the .hdr skies) are replaced by stated stand-ins: the
# procedural tower for eifel.obj, stellatedDode.obj for christ.obj, the synthetic sky; everything else (positions,
# materials, ids, light parameters, transforms) is the factory's own.  Cubes are omitted everywhere: the reference
# never intersects them (template/scene.h:1253-1255; bvh.cpp:20-22 counts triangles, spheres and planes only).

def _palette(b, rt):
    """The material set most factories open with (template/scene.h:815-829, :845-856, :1009-1020)."""
    m = {}
    m["blueDiff"] = b.diffuse(0.8, BLUE, 0.2, 0.8, 1, rt=rt)
    m["redDiff"] = b.diffuse(0.8, RED, 0.2, 0.8, 1, rt=rt)
    m["whiteDiff"] = b.diffuse(0.8, WHITE, 0.0, 1.0, 4, rt=rt)
    m["greenDiff"] = b.diffuse(0.8, GREEN, 0.2, 0.8, 2, rt=rt)
    m["standardGlass"] = b.glass(1.5, WHITE, rt=rt)
    m["blueGlass"] = b.glass(1.5, BABYBLUE, rt=rt)
    m["blueMetal"] = b.metal(0.7, BLUE, rt=rt)
    m["standardMetal"] = b.metal(0.7, WHITE, rt=rt)
    m["greenMetal"] = b.metal(0.7, GREEN, rt=rt)
    m["redMetal"] = b.metal(0.7, RED, rt=rt)
    m["yellowMetal"] = b.metal(0.7, GOLD, rt=rt)
    m["pinkMetal"] = b.metal(0.7, PINK, rt=rt)
    return m


def _tower_mesh(b, group, mat, pos, scale):
    """Mesh(group, "Resources/eifel.obj", mat, pos, scale) with the tower stand-in (the factory's scale 0.08 suits the
    real model's units; the stand-in is ~6 units tall and is given the scale that keeps the world size)."""
    tris = assets.lattice_tower(60, 24).reshape(-1, 3)  # 11,520 triangles
    tris = (tris * np.float32(scale) + np.array(pos, dtype=np.float32)).astype(np.float32).reshape(-1, 9)
    return b.mesh_raw(group, mat, tris)


def pretty_scene1(b, rt=True, split=BINNEDSAH):
    """instantiatePrettyScene1 (template/scene.h:814-840): unity.tri in the scene BVH (config 3 instances it instead)."""
    b.sky(assets.synthetic_sky(seed=5))
    b.area_light(11, (0.1, 4.0, 5.0), 8.0, WHITE, 1.0, (0, -1, 0))
    b.area_light(12, (0.1, 4.0, 3.0), 10.0, WHITE, 1.0, (0, -1, 0))
    light_diff = b.diffuse(0.8, WHITE, 0.6, 0.4, 1200, emission=1.2, rt=rt)
    standard_glass = b.glass(1.5, WHITE, rt=rt)
    blue_diff = b.diffuse(0.8, BLUE, 0.6, 0.4, 10, rt=rt)
    blue_glass = b.glass(1.5, BABYBLUE, rt=rt)
    yellow_metal = b.metal(0.7, GOLD, rt=rt)
    pink_metal = b.metal(0.7, PINK, rt=rt)
    b.sphere(7, blue_diff, (-0.7, -0.5, 2.0), 0.5)
    b.sphere(8, blue_glass, (-1.9, -0.5, 2.0), 0.5)
    b.sphere(9, yellow_metal, (-3.1, -0.5, 2.0), 0.5)
    b.sphere(6, pink_metal, (-4.3, -0.5, 2.0), 0.5)
    b.plane(1, light_diff, (0, 1, 0), 1)
    b.mesh_tri(2, assets.tri_path("unity"), standard_glass)
    b.build(split)
    return dict(name="pretty_scene1", tlas=False)


def pretty_animation_scene(b, rt=True, split=BINNEDSAH):
    """instantiatePrettyAnimationScene (:842-868): BigB.obj between four walls, the scene SetTime animates."""
    b.sky(assets.synthetic_sky(seed=6))
    m = _palette(b, rt)
    b.area_light(11, (-1, 8.0, -1), 12.0, WHITE, 2.0, (0, -1, 0))
    b.plane(0, m["whiteDiff"], (0, 1, 0), 1)
    b.plane(4, m["blueDiff"], (0, 0, -1), 10)
    b.plane(3, m["blueDiff"], (0, 0, 1), 10)
    b.plane(1, m["greenDiff"], (-1, 0, 0), 2.99)
    b.mesh_obj(2, assets.obj_path("BigB"), m["redDiff"], (-2, 1.0, 0), 4.0)
    b.build(split)
    return dict(name="pretty_animation", tlas=False)


def bigb_scene(b, rt=True, split=BINNEDSAH):
    """instantiateBigBScene (:880-889)."""
    b.sky(assets.synthetic_sky(seed=7))
    light_diff = b.diffuse(0.8, WHITE, 0.6, 0.4, 1200, emission=1.2, rt=rt)
    b.plane(2, light_diff, (0, 1, 0), 1)
    b.area_light(11, (0.1, 7.0, 3.0), 8.0, WHITE, 1.0, (0, -1, 0))
    blue_diff = b.diffuse(0.8, BLUE, 0.2, 0.8, 1, rt=rt)
    b.mesh_obj(2, assets.obj_path("BigB"), blue_diff, (0, 2.3, 5.0), 6.0)
    b.build(split)
    return dict(name="bigb_scene", tlas=False)


def christ_scene(b, rt=True, split=BINNEDSAH):
    """instantiateChristScene (:891-899) / instantiateScene8 (:1200-1208: the same mesh, another light); christ.obj is
    missing from the snapshot: stellatedDode.obj stands in, scaled to a comparable size."""
    b.sky(assets.synthetic_sky(seed=8))
    light_diff = b.diffuse(0.8, WHITE, 0.6, 0.4, 1200, emission=1.2, rt=rt)
    b.plane(2, light_diff, (0, 1, 0), 1)
    b.area_light(11, (0.1, 7.0, 5.0), 8.0, WHITE, 1.0, (0, -1, 0))
    yellow_metal = b.metal(0.7, GOLD, rt=rt)
    b.mesh_obj(2, assets.obj_path("stellatedDode"), yellow_metal, (0, 0.6, 6.2), 1.5)
    b.build(split)
    return dict(name="christ_scene", tlas=False)


def tlas_test(b, rt=True, split=BINNEDSAH):
    """TLASSceneTest (:901-939): three DIFFERENT meshes, one instance each (BigB.obj; christ.obj and eifel.obj by their
    stand-ins, with scales that give the factory's world sizes)."""
    b.sky(assets.synthetic_sky(seed=3))
    standard_metal = b.metal(0.7, WHITE, rt=rt)
    gold_metal = b.metal(0.7, GOLD, rt=rt)
    gold_diff = b.diffuse(0.8, GOLD, 0.8, 0.2, 1, rt=rt)
    red_diff = b.diffuse(0.8, RED, 0.0, 1, 1, rt=rt)
    spec_refl = b.diffuse(0.7, WHITE, 0.6, 0.4, 50, rt=rt)
    b.area_light(11, (0, 8.0, 0), 10.0, WHITE, 1.0, (0, -1, 0))
    m0 = b.mesh_obj(1, assets.obj_path("BigB"), red_diff, (0, 0.5, 0), 1)
    m1 = b.mesh_obj(2, assets.obj_path("stellatedDode"), gold_metal, (0, 0.5, 0), 1)
    m2 = _tower_mesh(b, 3, standard_metal, (0, 0.5, 0), 1)
    T0 = b.trs((0, 0, 3), 4, 0.0, _f32(PI_F * 0.5), 0.0)
    T1 = b.trs((2, 0, 3), 1.0, 0.0, PI_F, 0.0)
    T2 = b.trs((_f32(-2.3), 0, 3), 0.5, 0.0, _f32(PI_F * 0.5), 0.0)
    b.plane(0, spec_refl, (0, 1, 0), 0)
    b.sphere(7, gold_diff, (1.8, -0.5, 2.0), 0.5)
    b.build_tlas(split, [(m0, T0), (m1, T1), (m2, T2)])
    return dict(name="tlas_test", tlas=True)


def scene1(b, rt=True, split=BINNEDSAH):
    """instantiateScene1 (:974-1004), animation off: the box of six planes, glass ball, "rounded corners" sphere, ico.obj."""
    b.sky(assets.synthetic_sky(seed=12))
    standard_glass = b.glass(1.5, WHITE, rt=rt)
    specular_diff = b.diffuse(0.8, WHITE, 0.6, 0.4, 2, rt=rt)
    light_diff = b.diffuse(0.8, WHITE, 0.6, 0.4, 1200, emission=1.2, rt=rt)
    green_diff = b.diffuse(0.8, GREEN, 0.6, 0.4, 2, rt=rt)
    red_diff = b.diffuse(0.8, RED, 0.6, 0.4, 2, rt=rt)
    spec_refl = b.diffuse(0.7, WHITE, 0.6, 0.4, 50, rt=rt)
    b.area_light(11, (0.1, 1.95, 1.5), 4.0, WHITE, 1.0, (0, -1, 0))
    b.plane(0, red_diff, (1, 0, 0), 3)
    b.plane(1, green_diff, (-1, 0, 0), 2.99)
    b.plane(2, spec_refl, (0, 1, 0), 1)
    b.plane(3, light_diff, (0, -1, 0), 2)
    b.plane(4, light_diff, (0, 0, 1), 3)
    b.plane(5, specular_diff, (0, 0, -1), 3.99)
    b.sphere(7, standard_glass, (-1.5, -0.5, 2), 0.5)
    b.sphere(8, b.diffuse(0.8, WHITE, 0, 0.3, 0, rt=rt), (0, 2.5, -3.07), 8)  # diffuse(0.8f, white, 0, 0.3f, 0.7f -> int 0, rt)
    b.mesh_obj(1, assets.obj_path("ico"), green_diff, (0.1, -0.6, 1.5), 0.5)
    b.build(split)
    return dict(name="scene1", tlas=False)


def scene2(b, rt=True, split=BINNEDSAH):
    """instantiateScene2 (:1006-1043): eight spheres of every material, four planes, the tower (eifel.obj stand-in)."""
    b.sky(assets.synthetic_sky(seed=13))
    m = _palette(b, rt)
    b.area_light(11, (-1, 8.0, -1), 25.0, WHITE, 3.0, (0, -1, 0))
    b.plane(0, m["whiteDiff"], (0, 1, 0), 1)
    b.plane(4, m["blueDiff"], (0, 0, -1), 10)
    b.plane(3, m["blueDiff"], (0, 0, 1), 10)
    b.plane(1, m["greenDiff"], (-1, 0, 0), 2.99)
    for idx, mat, pos in ((7, "redDiff", (-0.7, -0.5, 2.0)), (8, "blueGlass", (-1.9, -0.5, 2.0)), (9, "yellowMetal", (-3.1, -0.5, 2.0)),
                          (6, "pinkMetal", (-4.3, -0.5, 2.0)), (10, "greenMetal", (-0.7, -0.5, 3.2)), (13, "standardGlass", (-1.9, -0.5, 3.2)),
                          (14, "redDiff", (-3.1, -0.5, 3.2)), (15, "redMetal", (-4.3, -0.5, 3.2))):
        b.sphere(idx, m[mat], pos, 0.5)
    _tower_mesh(b, 2, m["standardMetal"], (-2, -1.0, 0), 0.8)
    b.build(split)
    return dict(name="scene2", tlas=False)


def scene4(b, rt=True, split=BINNEDSAH):
    """instantiateScene4 (:1094-1115); the light's normal points UP in the factory."""
    b.sky(assets.synthetic_sky(seed=14))
    blue_glass = b.glass(1.5, BABYBLUE, rt=rt)
    standard_metal = b.metal(0.7, WHITE, rt=rt)
    light_diff = b.diffuse(0.8, PINK, 0.6, 0.4, 30, rt=rt)
    gold_diff = b.diffuse(0.8, GOLD, 0.6, 0.4, 30, rt=rt)
    b.area_light(11, (1.8, 2.0, 5.5), 10.0, WHITE, 2.0, (0, 1, 0))
    b.plane(0, b.diffuse(0.8, WHITE, 0.0, 1.0, 4, rt=rt), (0, 1, 0), 1)
    b.sphere(7, blue_glass, (-0.7, -0.5, 2.0), 0.5)
    b.sphere(8, standard_metal, (-2.2, -0.5, 2.0), 0.5)
    b.sphere(9, light_diff, (-3.7, -0.5, 2.0), 0.5)
    b.sphere(5, gold_diff, (1.8, -0.5, 2.0), 0.5)
    b.mesh_obj(1, assets.obj_path("ico"), standard_metal, (0.5, -0.51, 2), 0.5)
    b.build(split)
    return dict(name="scene4", tlas=False)


def scene5(b, rt=True, split=BINNEDSAH, mesh="BigB"):
    """instantiateScene5 (:1117-1131): BigB.obj in the scene BVH, two lights."""
    b.sky(assets.synthetic_sky(seed=15))
    gold_diff = b.diffuse(0.8, GOLD, 0.6, 0.4, 30, rt=rt)
    b.area_light(11, (1, 2.0, 1), 10.0, WHITE, 1.0, (0, -1, 0))
    b.area_light(12, (-1, 2.0, -1), 5.0, WHITE, 1.0, (0, -1, 0))
    b.plane(0, b.diffuse(0.8, WHITE, 0.0, 1.0, 4, rt=rt), (0, 1, 0), 0)
    b.mesh_obj(1, assets.obj_path(mesh), gold_diff, (0, 0.5, 0), 1)
    b.build(split)
    return dict(name="scene5", tlas=False)


def scene6(b, rt=True, split=BINNEDSAH):
    """instantiateScene6 (:1133-1170): lowBigB.obj, instanced.  The factory fills four transforms into arrays of
    bvhCount = 3 (template/scene.h:1372; the fourth write is out of bounds) and the TLAS is built over three."""
    b.sky(assets.synthetic_sky(seed=16))
    gold_diff = b.diffuse(0.8, GOLD, 0.8, 0.2, 1, rt=rt)
    red_diff = b.diffuse(0.8, RED, 0.0, 1, 1, rt=rt)
    b.area_light(11, (0, 6.0, 0), 16.0, WHITE, 2.0, (0, -1, 0))
    m = b.mesh_obj(1, assets.obj_path("lowBigB"), red_diff, (0, 0.5, 0), 1)
    ry = _f32(PI_F * 0.5)
    inst = [(m, b.trs((0, 0, 3), 4, 0.0, ry, 0.0)), (m, b.trs((-3, 0, 2), 4, 0.0, ry, 0.0)), (m, b.trs((2, 0, -3), 4, 0.0, ry, 0.0))]
    b.plane(0, b.diffuse(0.8, WHITE, 0.0, 1.0, 4, rt=rt), (0, 1, 0), 0)
    b.sphere(7, gold_diff, (1.8, -0.5, 2.0), 0.5)
    b.build_tlas(split, inst)
    return dict(name="scene6", tlas=True)


def _xorshift32_floats(seed=0x12345678):
    """RandomFloat() of the template (template/template.cpp:684-691) from its initial seed, as a generator."""
    s = seed
    while True:
        s ^= (s << 13) & 0xFFFFFFFF
        s ^= s >> 17
        s ^= (s << 5) & 0xFFFFFFFF
        yield float(np.float32(s) * np.float32(2.3283064365387e-10))


def scene7(b, rt=True, split=BINNEDSAH, nx=255, ny=255):
    """instantiateScene7 (:1172-1198): nx x ny unit spheres on the floor, material and colour of each drawn from the
    template's global RandomFloat() stream (assumed untouched before the Scene is constructed: the Scene is a member
    of the Renderer, built before anything else draws); ids 13 + x + y collide by design."""
    b.sky(assets.synthetic_sky(seed=17))
    white_diff = b.diffuse(0.8, WHITE, 0.0, 1.0, 4, rt=rt)
    b.plane(0, white_diff, (0, 1, 0), 1)
    b.area_light(11, (-1, 8.0, -1), 25.0, WHITE, 3.0, (0, -1, 0))
    rnd = _xorshift32_floats()
    for x in range(nx):
        for y in range(ny):
            die = next(rnd)
            col = (next(rnd), next(rnd), next(rnd))  # float3(RandomFloat(), RandomFloat(), RandomFloat()): argument order x, y, z
            if die < 0.33:
                mat = b.metal(0.7, col, rt=rt)
            elif die < 0.66:
                mat = b.glass(1.5, col, rt=rt)
            else:
                mat = b.diffuse(0.8, col, 0.2, 0.8, 2, rt=rt)
            b.sphere(13 + x + y, mat, (float(x), -0.5, float(y)), 0.5)
    b.build(split)
    return dict(name="scene7", tlas=False)


REGISTRY = {
    "terrain": terrain_scene,
    "qlearn_probe": qlearn_probe,
    "config1": config1, "config2": config2, "config3": config3, "config4": config4, "config5": config5,
    "background": background_scene, "scene3": scene3, "pretty_tlas": pretty_tlas, "tower": tower_scene,
    "bigb_instanced": bigb_instanced, "tlas_test2": tlas_test2, "mixed_small": mixed_small,
    # every factory of template/scene.h:791-1209 (instantiateBackgroundScene = "background", instantiateScene3 = "scene3",
    # instantiateEifelScene = "tower", TLASSceneTest2 = "tlas_test2", instantiateScene8 = "christ_scene")
    "pretty_scene1": pretty_scene1, "pretty_animation": pretty_animation_scene, "bigb_scene": bigb_scene, "christ_scene": christ_scene,
    "tlas_test": tlas_test, "scene1": scene1, "scene2": scene2, "scene4": scene4, "scene5": scene5, "scene6": scene6, "scene7": scene7,
}
REFERENCE_FACTORIES = {  # reference factory -> registry name
    "instantiateBackgroundScene": "background", "instantiatePrettyScene1": "pretty_scene1", "instantiatePrettyAnimationScene": "pretty_animation",
    "instantiateEifelScene": "tower", "instantiateBigBScene": "bigb_scene", "instantiateChristScene": "christ_scene", "TLASSceneTest": "tlas_test",
    "TLASSceneTest2": "tlas_test2", "instantiateScene1": "scene1", "instantiateScene2": "scene2", "instantiateScene3": "scene3",
    "instantiateScene4": "scene4", "instantiateScene5": "scene5", "instantiateScene6": "scene6", "instantiateScene7": "scene7",
    "instantiateScene8": "christ_scene",
}
