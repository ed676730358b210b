import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Native pieces compiled in-tree (no-op when up to date)."""
    import __graft_entry__ as entry
    entry.build()
    return True


@pytest.fixture(scope="session")
def small_textures():
    from raytracing_opengl_amd import textures
    return textures.default_texture_set(scale=16)


@pytest.fixture(scope="session")
def mid_textures():
    from raytracing_opengl_amd import textures
    return textures.default_texture_set(scale=4)
