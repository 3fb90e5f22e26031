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


def pytest_collection_modifyitems(config, items):
    # A GPU test that hangs (a box whose GPU stops answering: DESIGN.md 0g) must end as a failure with a stack dump, by itself, so
    # that whatever runs after the suite still gets its turn.  The whole GPU suite takes about seven minutes; ten per test is far
    # above any of them.  Tests that carry their own timeout mark keep it; CPU tests are left alone.
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for it in items:
        if it.get_closest_marker("gpu") is not None and it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(int(os.environ.get("KC_GPU_TEST_TIMEOUT", "600")), method="thread"))


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
