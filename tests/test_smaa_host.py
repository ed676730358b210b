"""The product's SMAA arithmetic (csrc/smaa_device.h), compiled for the host, against the independently written oracle: every byte
of the edge, weight and screen textures must be IDENTICAL, for all four presets, on traced frames, synthetic patterns (long lines
into the border, every slope, checkerboards, thresholds) and odd sizes. The GPU tests (tests/test_gpu_smaa.py) then check the
kernels' plumbing against the same oracle."""
import numpy as np
import pytest

import harness
import smaa_cases
import smaa_tables
from oracle import smaa


@pytest.fixture(scope="module")
def tables():
    return smaa_tables.area_table(), smaa_tables.search_table()


def _same(a, b):
    for k in ("edges", "blend", "screen"):
        assert np.array_equal(a[k], b[k]), (k, int((a[k] != b[k]).sum()))


@pytest.mark.parametrize("preset", smaa.PRESETS)
@pytest.mark.parametrize("seed,w,h", [(1, 320, 200), (2, 203, 131), (3, 64, 16), (4, 5, 3), (5, 1, 1), (6, 131, 77)])
def test_patterns_byte_exact(built, tables, preset, seed, w, h):
    img = smaa_cases.pattern(seed, w, h)
    _same(harness.smaa(img, preset, *tables), smaa.run(img, preset, *tables))


@pytest.mark.parametrize("kind,w,h,depth,preset", [("default", 256, 144, 4, "ULTRA"), ("torus", 224, 126, 6, "HIGH"), ("quadric", 224, 126, 4, "LOW")])
def test_traced_frames_byte_exact(built, tables, kind, w, h, depth, preset):
    img = smaa_cases.traced(kind, w, h, depth)
    _same(harness.smaa(img, preset, *tables), smaa.run(img, preset, *tables))


def test_random_tables_and_noise_byte_exact(built):
    """Arbitrary table contents and pure noise frames: no structure to hide an indexing error behind."""
    rng = np.random.default_rng(9)
    for k in range(4):
        area = rng.integers(0, 256, smaa.AREA_SHAPE, dtype=np.uint8)
        search = rng.choice(np.array([0, 127, 254], np.uint8), smaa.SEARCH_SHAPE)
        img = rng.integers(0, 256, (48, 80, 4), dtype=np.uint8)
        img[..., :3] = (img[..., :3] // 64) * 64            # coarse levels: plenty of long runs and crossings
        _same(harness.smaa(img, smaa.PRESETS[k], area, search), smaa.run(img, smaa.PRESETS[k], area, search))


def test_divide_free_unorm8_is_exact(built):
    import ctypes
    lib = harness.lib()
    lib.harness_smaa_unorm8_mismatches.restype = ctypes.c_int
    assert lib.harness_smaa_unorm8_mismatches() == 0
