import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """libirx.so, built in-tree if stale (hipcc cross-compiles gfx950 without a GPU)."""
    from instancerefer_amd import _build, _lib
    _build.build_lib()
    return _lib.load()
