"""Mesh fixtures: the reference's Resources/*.obj and unity.tri as numbers (assets/*.npz, packed by
tests/golden/make_assets.py), re-emitted as text files in the two dialects the reference's Mesh
loaders read (template/scene.h:261-313) so the host loaders parse real files.

'%.9g' round-trips float32 exactly, so a loader using strtof semantics recovers the packed values.
"""
import os
import tempfile
import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
_CACHE = None


def _cache_dir():
    global _CACHE
    if _CACHE is None:
        _CACHE = tempfile.mkdtemp(prefix="rapt_assets_")
    return _CACHE


def obj_arrays(name):
    z = np.load(os.path.join(_DIR, name + "_obj.npz"))
    return z["v"], z["f"]


def tri_rows(name):
    return np.load(os.path.join(_DIR, name + "_tri.npz"))["rows"]


def write_obj(path, v, f):
    with open(path, "w") as fh:
        for p in v:
            fh.write("v %.9g %.9g %.9g\n" % (p[0], p[1], p[2]))
        for t in f:
            fh.write("f %d//%d %d//%d %d//%d\n" % (t[0], t[0], t[1], t[1], t[2], t[2]))


def write_tri(path, rows, trailing_newline=False):
    # unity.tri ends without a newline after its sentinel record; keep that shape
    with open(path, "w") as fh:
        fh.write("\n".join(" ".join("%.9g" % x for x in r) for r in rows))
        if trailing_newline:
            fh.write("\n")


def obj_path(name):
    """Path of a text .obj file for fixture `name` (ico, three, stellatedDode, lowBigB, BigB)."""
    p = os.path.join(_cache_dir(), name + ".obj")
    if not os.path.exists(p):
        v, f = obj_arrays(name)
        write_obj(p, v, f)
    return p


def tri_path(name):
    """Path of a text .tri file for fixture `name` (unity)."""
    p = os.path.join(_cache_dir(), name + ".tri")
    if not os.path.exists(p):
        write_tri(p, tri_rows(name))
    return p


def synthetic_sky(width=512, height=256, seed=7):
    """Seeded 8-bit lat-long sky texture standing in for the missing Resources/*.hdr files
    (.MISSING_LARGE_BLOBS): vertical gradient + sun blob + low-frequency noise, RGB uint8."""
    rng = np.random.default_rng(seed)
    y = np.linspace(0.0, 1.0, height, dtype=np.float32)[:, None]
    x = np.linspace(0.0, 1.0, width, dtype=np.float32)[None, :]
    top = np.array([70, 120, 220], dtype=np.float32)
    hor = np.array([225, 225, 235], dtype=np.float32)
    gnd = np.array([90, 80, 70], dtype=np.float32)
    img = np.where(y[..., None] < 0.5, top + (hor - top) * (y[..., None] * 2), hor + (gnd - hor) * ((y[..., None] - 0.5) * 2))
    img = np.broadcast_to(img, (height, width, 3)).copy()
    sun = np.exp(-(((x - 0.7) * 6) ** 2 + ((y - 0.25) * 8) ** 2))
    img += sun[..., None] * 60
    noise = rng.normal(0, 1, (height // 16 + 1, width // 16 + 1)).astype(np.float32)
    noise = np.kron(noise, np.ones((16, 16), dtype=np.float32))[:height, :width]
    img += noise[..., None] * 6
    return np.clip(img, 0, 255).astype(np.uint8)


def lattice_tower(levels=160, sides=40, seed=3):
    """Procedural stand-in for the missing eifel.obj: a tapering lattice tower of thin triangular
    struts.  Returns float32 [n, 9] triangles; n = levels * sides * 8 = 51,200 by default (SURVEY.md
    section 8d asks for a stand-in of 50-200 k triangles)."""
    rng = np.random.default_rng(seed)
    tris = []
    def ring(k):
        h = k / levels
        r = 1.2 * (1 - h) ** 1.6 + 0.05
        a = np.arange(sides) * (2 * np.pi / sides) + 0.15 * k
        return np.stack([r * np.cos(a), np.full(sides, 6.0 * h), r * np.sin(a)], axis=1)
    def strut(p, q, w):
        d = q - p
        n = np.cross(d, rng.normal(size=3))
        n = n / (np.linalg.norm(n) + 1e-9) * w
        tris.append(np.concatenate([p - n, p + n, q + n]))
        tris.append(np.concatenate([p - n, q + n, q - n]))
    for k in range(levels):
        a, b = ring(k), ring(k + 1)
        for s in range(sides):
            s2 = (s + 1) % sides
            strut(a[s], b[s], 0.02)
            strut(a[s], a[s2], 0.02)
            strut(a[s], b[s2], 0.015)
            strut(a[s2], b[s], 0.015)
    return np.array(tris, dtype=np.float32)


def write_hdr(path, rgb, rle=True):
    """Write a Radiance RGBE (.hdr) file from float RGB [H, W, 3] -- the published format: header,
    '-Y h +X w', then flat RGBE pixels or new-style run-length-encoded scanlines (per channel)."""
    rgb = np.asarray(rgb, dtype=np.float32)
    h, w, _ = rgb.shape
    m = rgb.max(axis=2)
    e = np.zeros_like(m, dtype=np.int32)
    nz = m > 1e-32
    mant, ex = np.frexp(m[nz])
    scale = np.zeros_like(m)
    scale[nz] = mant * 256.0 / m[nz]
    e[nz] = ex + 128
    rgbe = np.zeros((h, w, 4), dtype=np.uint8)
    rgbe[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.clip(e, 0, 255).astype(np.uint8)
    rgbe[~nz] = 0
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        if not rle or w < 8 or w >= 32768:
            f.write(rgbe.tobytes())
            return rgbe
        for j in range(h):
            f.write(bytes([2, 2, w >> 8, w & 255]))
            for k in range(4):
                row = rgbe[j, :, k]
                i = 0
                while i < w:
                    run = 1
                    while i + run < w and run < 127 and row[i + run] == row[i]:
                        run += 1
                    if run >= 4:
                        f.write(bytes([128 + run, int(row[i])]))
                        i += run
                    else:
                        lit = i
                        while lit < w and lit - i < 128:
                            r2 = 1
                            while lit + r2 < w and r2 < 4 and row[lit + r2] == row[lit]:
                                r2 += 1
                            if r2 >= 4:
                                break
                            lit += 1
                        n = max(1, lit - i)
                        f.write(bytes([n]) + row[i:i + n].tobytes())
                        i += n
    return rgbe


def synthetic_sky_hdr(width=256, height=128, seed=7):
    """Float lat-long sky (linear radiance, sun well above 1.0) for .hdr round trips."""
    base = synthetic_sky(width, height, seed).astype(np.float32) / 255.0
    lin = base ** 2.2
    y = np.linspace(0.0, 1.0, height, dtype=np.float32)[:, None]
    x = np.linspace(0.0, 1.0, width, dtype=np.float32)[None, :]
    sun = np.exp(-(((x - 0.7) * 20) ** 2 + ((y - 0.25) * 24) ** 2))
    lin += sun[..., None] * 40.0
    lin[height - 4:, :8] = 0.0  # exact zeros (e = 0 texels)
    return lin


def terrain(n=2048, seed=11, size=64.0, height=6.0):
    """A seeded procedural heightfield of n x n cells = 2 n^2 triangles (n = 2048: 8,388,608), (T, 9) float32: five octaves of
    value noise (bilinear, seeded) over a size x size square centred on the origin, heights in [0, height].  The out-of-cache
    stand-in for the reference's big meshes (eifel.obj, christ.obj: .MISSING_LARGE_BLOBS): at n = 2048 its pair and
    primitive records are 1 GiB, four times the MI355X's Infinity Cache (profiles/out_of_cache.py)."""
    rng = np.random.default_rng(seed)
    v = n + 1
    u = np.linspace(0.0, 1.0, v, dtype=np.float64)
    h = np.zeros((v, v), np.float64)
    amp, freq = 1.0, 4
    for _ in range(5):
        g = rng.random((freq + 2, freq + 2))
        x = u * freq
        i0 = np.minimum(x.astype(np.int64), freq)
        f = x - i0
        f = f * f * (3 - 2 * f)
        row = g[i0][:, None, :] * (1 - f)[:, None, None] + g[i0 + 1][:, None, :] * f[:, None, None]  # [v, 1, freq + 2]
        a = row[:, 0, :][:, i0] * (1 - f)[None, :] + row[:, 0, :][:, i0 + 1] * f[None, :]
        h += amp * a
        amp *= 0.5
        freq *= 2
    h = (h - h.min()) / (h.max() - h.min()) * height
    xs = ((u - 0.5) * size).astype(np.float32)
    X, Z = np.meshgrid(xs, xs, indexing="ij")
    P = np.stack([X, h.astype(np.float32), Z], -1)  # [v, v, 3]
    p00, p10, p01, p11 = P[:-1, :-1], P[1:, :-1], P[:-1, 1:], P[1:, 1:]
    tris = np.empty((n, n, 2, 9), np.float32)
    tris[:, :, 0, 0:3], tris[:, :, 0, 3:6], tris[:, :, 0, 6:9] = p00, p01, p10  # counter-clockwise seen from above
    tris[:, :, 1, 0:3], tris[:, :, 1, 3:6], tris[:, :, 1, 6:9] = p10, p01, p11
    return tris.reshape(-1, 9)
