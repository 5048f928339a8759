import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    p = graft.load_package()
    # built libraries travel with the snapshot; only (re)build when something is missing
    if not (os.path.exists(p.HIP_LIB) and os.path.exists(p.HOST_LIB)):
        p.build()
    return p


@pytest.fixture(scope="session")
def orc():
    o = graft.load_oracle()
    o.build()
    return o


@pytest.fixture(scope="session")
def gpu_renderer(pkg):
    import torch  # noqa: F401  (first, so the HIP library shares torch's HIP runtime)

    r = pkg.Renderer(device=0)
    yield r
    r.close()
