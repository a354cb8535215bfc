import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_cases():
    out = []
    for fn in sorted(os.listdir(GOLDEN)):
        if fn.endswith(("_pt.npz", "_dl.npz")):
            out.append(fn)
    return out


def load_golden(fn):
    z = np.load(os.path.join(GOLDEN, fn), allow_pickle=False)
    return dict(scene=str(z["scene"]), w=int(z["width"]), h=int(z["height"]), spp=int(z["spp"]),
                integrator=int(z["integrator"]), colors=z["colors"], pixels=z["pixels"],
                seeds_after=z["seeds_after"])


def same_bits(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.tobytes() == b.tobytes()


@pytest.fixture(scope="session")
def have_ref():
    import oracle_binding as ob
    return ob.ref_available()
