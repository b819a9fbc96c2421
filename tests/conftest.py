import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(name=""):
    """Import a module of the product package (its directory name has a hyphen)."""
    return importlib.import_module("ray-and-pathtracer_amd" + ("." + name if name else ""))


@pytest.fixture(scope="session")
def scenes():
    return pkg("scenes")


@pytest.fixture(scope="session")
def host_api():
    m = pkg("host_api")
    m.build()
    return m


@pytest.fixture(scope="session")
def oracle_api():
    from oracle import oracle_api as oa
    oa.build()
    return oa


def rel_err(a, b, floor=1e-3):
    """Relative error |a-b| / max(|b|, floor) on finite entries; non-finite entries must agree by class."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    fa, fb = np.isfinite(a), np.isfinite(b)
    cls_ok = np.array_equal(fa, fb) and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isposinf(a), np.isposinf(b))
    m = fa & fb
    err = np.zeros_like(a)
    err[m] = np.abs(a[m] - b[m]) / np.maximum(np.abs(b[m]), floor)
    return err, cls_ok


def random_rays(n, seed, center=(0.5, 0.8, 1.0), spread=4.0):
    """Seeded rays: origins in a box around the scenes' content, unit directions, some axis-aligned
    (zero components make rD infinite, which exercises the NaN behaviour of the slab test)."""
    rng = np.random.default_rng(seed)
    O = (rng.uniform(-1, 1, (n, 3)) * spread + np.array(center)).astype(np.float32)
    D = rng.normal(size=(n, 3)).astype(np.float32)
    k = n // 16
    D[:k, 0] = 0
    D[k:2 * k, 1] = 0
    D[2 * k:3 * k, 2] = 0
    D[3 * k:3 * k + 8] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [0, -1, 0], [0, 0, 1]], dtype=np.float32)[: max(0, min(8, n - 3 * k))]
    nrm = np.sqrt((D.astype(np.float64) ** 2).sum(1, keepdims=True))
    D = (D / np.maximum(nrm, 1e-20)).astype(np.float32)
    return O, D
