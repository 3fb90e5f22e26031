import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def kclib():
    """The product library (built in-tree).  Build it if missing — hipcc cross-compiles without a GPU."""
    from compress_amd import _lib
    if not os.path.exists(_lib.lib_path()):
        from compress_amd import build
        build.build()
    return _lib.load()
