import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    p = ge.load_package()
    if not os.path.exists(os.path.join(ROOT, "pop2-cesm_amd", "libpop_amd.so")):
        p.build()
    return p


@pytest.fixture(scope="session")
def orclib_built():
    import orclib
    orclib.build()
    return orclib
