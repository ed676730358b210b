import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _cpu_budget():
    """OpenMP's default team is one thread per VISIBLE hardware thread; the GPU boxes show 256 and grant a container 16 CPUs of time (cgroup
    quota), where 256 threads are slower than 16 (tools/time_oracle_threads.py). The checkers' OpenMP loops (oracle/*.c, tests/host_harness)
    get the CPUs this container may actually use, unless the caller has set OMP_NUM_THREADS."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.999)))
    except Exception:
        pass
    return n


os.environ.setdefault("OMP_NUM_THREADS", str(_cpu_budget()))     # before any OpenMP library is loaded


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
