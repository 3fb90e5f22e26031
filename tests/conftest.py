import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _no_core_dumps():
    # A faulting GPU process dumps its whole address space (tens of GiB with a 4 GiB batch resident): on a box with a small
    # scratch disk that fills it and takes the box down with the test.  The suite never reads a core file.
    try:
        import resource
        resource.setrlimit(resource.RLIMIT_CORE, (0, 0))
    except Exception:
        pass


def pytest_configure(config):
    _no_core_dumps()
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
