"""The host's Radiance .hdr -> 8-bit skydome loader (Scene::LoadSkyHDR, the stbi_load(..., 3) call of
template/scene.h:792) against the reference's OWN vendored stb_image.h, built from where it lies
(oracle/ref -> oracle/_ref/libstb_ref.so).  This is the one input of the path for which the real
reference code builds in this image without stand-ins, so here parity is pinned, bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libstb_ref.so")


def stb_load(path):
    L = C.CDLL(REF_SO)
    L.ref_stbi_load.restype = C.POINTER(C.c_ubyte)
    w, h, n = C.c_int(), C.c_int(), C.c_int()
    p = L.ref_stbi_load(path.encode(), C.byref(w), C.byref(h), C.byref(n), 3)
    assert p, "stb_image could not load " + path
    out = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    L.ref_stbi_free(p)
    return out, n.value


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("w,h,rle", [(256, 128, True), (256, 128, False), (64, 33, True), (7, 5, True), (9, 3, False)])
def test_hdr_loader_matches_reference_stb_image(w, h, rle, tmp_path, host_api):
    from conftest import pkg
    assets = pkg("assets")
    path = str(tmp_path / "sky.hdr")
    assets.write_hdr(path, assets.synthetic_sky_hdr(w, h, seed=w + h), rle=rle)
    ref, comps = stb_load(path)
    s = host_api.HostScene()
    got = s.sky_hdr(path)
    assert comps == 3 and got.shape == ref.shape
    assert np.array_equal(got, ref)
    if w >= 64:
        assert len(np.unique(ref)) > 20 and ref.max() == 255 and ref.min() == 0  # the image exercises the range
    s.close()


def test_hdr_loader_rejects_bad_files(tmp_path, host_api):
    s = host_api.HostScene()
    bad = tmp_path / "bad.hdr"
    bad.write_bytes(b"P6\n1 1\n255\n\0\0\0")
    with pytest.raises(RuntimeError, match="Radiance"):
        s.sky_hdr(str(bad))
    trunc = tmp_path / "trunc.hdr"
    trunc.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 4 +X 16\n\2\2\0\20\x90")
    with pytest.raises(RuntimeError):
        s.sky_hdr(str(trunc))
    with pytest.raises(RuntimeError, match="open"):
        s.sky_hdr(str(tmp_path / "missing.hdr"))
    s.close()
