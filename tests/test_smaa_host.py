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


@pytest.mark.parametrize("planes", [True, False])
@pytest.mark.parametrize("preset", ["HIGH", "ULTRA", "MEDIUM"])
def test_weights_composed_from_independent_parts_are_the_same_bytes(built, tables, preset, planes):
    """smaa_weights_roles_kernel (round 4) gives the diagonal part -- once per pair of diagonals --, the north part and the west part of a
    pixel to four waves and applies the shader's selection rule to the results (smaa::BlendT::combine). The same composition on the host
    must give weights()' bytes: on patterns with every slope (diagonals of both orientations, crossing ones), on traced frames, odd sizes."""
    frames = [smaa_cases.pattern(seed, w, h) for seed, w, h in ((1, 320, 200), (2, 203, 131), (7, 97, 160), (4, 5, 3))]
    frames += [smaa_cases.traced("default", 256, 144, 4), smaa_cases.traced("torus", 224, 126, 6)]
    for img in frames:
        a = harness.smaa(img, preset, *tables, planes=planes, roles=True)
        _same(a, harness.smaa(img, preset, *tables, planes=planes))
        _same(a, smaa.run(img, preset, *tables))


def test_random_tables_and_noise_byte_exact(built):
    """Arbitrary table contents and pure noise frames: no structure to hide an indexing error behind."""
    rng = np.random.default_rng(9)
    for k in range(4):
        area = rng.integers(0, 256, smaa.AREA_SHAPE, dtype=np.uint8)
        search = rng.choice(np.array([0, 127, 254], np.uint8), smaa.SEARCH_SHAPE)
        img = rng.integers(0, 256, (48, 80, 4), dtype=np.uint8)
        img[..., :3] = (img[..., :3] // 64) * 64            # coarse levels: plenty of long runs and crossings
        _same(harness.smaa(img, smaa.PRESETS[k], area, search), smaa.run(img, smaa.PRESETS[k], area, search))


@pytest.mark.parametrize("preset", smaa.PRESETS)
def test_step_counts_from_the_bit_planes_change_nothing(built, tables, preset):
    """The HIP weight kernel computes how many steps an orthogonal search takes from whole words of two bit planes (smaa_device.h
    SearchPlanes) instead of walking the edge: with and without them the host build must produce the oracle's textures byte for byte -- on
    lines much longer than 2 x max_steps in both axes (full-length searches in all four directions), lines that end inside the window at
    every offset and parity, crossings, frame borders (where the counts repeat the border position), widths that are not multiples of 32 and
    heights that are not multiples of 8."""
    rng = np.random.default_rng(3)
    cases = [smaa_cases.pattern(7, 600, 90), smaa_cases.pattern(8, 97, 300)]
    long_lines = np.full((150, 700, 4), 255, np.uint8)
    long_lines[::37, :, :3] = 0
    long_lines[:, ::53, :3] = 0
    long_lines[70:, 300:, 1] = 90
    for k in range(140):
        long_lines[5 + k, 20 + k:23 + k, :3] = 30
    cases.append(long_lines)
    ends = np.full((260, 420, 4), 255, np.uint8)            # horizontal and vertical runs of every length 1..100 at shifting offsets
    for k in range(100):
        ends[2 * k + 3, 40 + (k % 7): 40 + (k % 7) + k + 1, :3] = 10
        ends[20 + (k % 5): 20 + (k % 5) + k + 1, 200 + 2 * k, :3] = 10
    cases.append(ends)
    noise = rng.integers(0, 256, (70, 300, 4), dtype=np.uint8)
    noise[..., :3] = (noise[..., :3] // 128) * 128
    cases.append(noise)
    # frames smaller than a search window: the window hangs over BOTH borders at once (the counts repeat the border position there, the way
    # CLAMP_TO_EDGE repeats the border texel), last plane words / blocks only partly inside the frame
    for hh, ww in ((9, 33), (7, 31), (17, 65), (64, 8), (1, 40), (40, 1), (3, 130)):
        small = rng.integers(0, 256, (hh, ww, 4), dtype=np.uint8)
        small[..., :3] = (small[..., :3] // 128) * 128
        cases.append(small)
        lines = np.full((hh, ww, 4), 255, np.uint8)
        lines[hh // 2, :, :3] = 0
        lines[:, ww // 2, :3] = 0
        lines[0, :, :3] = 40
        lines[:, ww - 1, :3] = 40
        cases.append(lines)
    for img in cases:
        want = smaa.run(img, preset, *tables)
        _same(harness.smaa(img, preset, *tables, planes=True), want)
        _same(harness.smaa(img, preset, *tables, planes=False), want)


def test_divide_free_unorm8_is_exact(built):
    import ctypes
    lib = harness.lib()
    lib.harness_smaa_unorm8_mismatches.restype = ctypes.c_int
    assert lib.harness_smaa_unorm8_mismatches() == 0
