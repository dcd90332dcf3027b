import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mp3-steganography-lib_amd")
for p in (PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def built():
    """Build the HIP extension and the oracle once per session (hipcc cross-compiles without a GPU)."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as G
    G.build()
    return True


@pytest.fixture(scope="session")
def orc(built):
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def mlib(built):
    from mp3stego import _lib
    _lib.lib()
    return _lib


@pytest.fixture(scope="session")
def ctx(mlib):
    c = mlib.Context(0)
    yield c
    c.close()
