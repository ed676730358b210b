"""The SMAA oracle (oracle/smaa_oracle.c) pinned against the reference's own SMAA shaders executed on Mesa llvmpipe.

tests/golden/smaa_ref_*.npz (tools/gen_smaa_fixtures.py) hold, per case, the input frame and the edge / weight / screen textures the
reference's three passes produced. Every pixel that differs is ACCOUNTED FOR by tests/smaa_classify.py: it is either within 1 LSB
(bilinear weight rounding) or it sits on one of the three decisions of SMAA.h that sub-texel float noise decides in a GL
implementation; anything else fails. Each pass is checked on the reference's OWN input of that pass, so that one flipped edge
cannot excuse the passes behind it, and the whole chain is checked end to end with a bound on how many pixels may differ."""
import glob
import os

import numpy as np
import pytest

import smaa_cases
import smaa_classify
import smaa_tables
from oracle import smaa

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = sorted(glob.glob(os.path.join(GOLD, "smaa_ref_*.npz")))


@pytest.fixture(scope="module")
def tables():
    return smaa_tables.area_table(), smaa_tables.search_table()


def test_fixture_set():
    assert len(FIXTURES) >= 8
    presets = {str(np.load(f)["preset"]) for f in FIXTURES}
    assert presets == set(smaa.PRESETS)


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(f)[9:-4] for f in FIXTURES])
def test_every_difference_from_the_reference_shaders_is_accounted_for(built, tables, path):
    fx = np.load(path)
    area, search = tables
    color, preset = fx["color"], str(fx["preset"])
    e = smaa_classify.classify_edges(color, fx["edges"], preset)
    assert e["unexplained"] == 0 and e["differing"] <= 3, e
    b = smaa_classify.classify_blend(fx["edges"], fx["blend"], preset, area, search)
    assert b["unexplained"] == 0, b
    n_edge = int((fx["edges"] != 0).any(-1).sum())
    assert b["differing"] <= 0.3 * n_edge, b             # the noise-decided pixels are a minority even among edge pixels (measured: <= 28 %)
    n = smaa_classify.classify_neighborhood(color, fx["blend"], fx["screen"])
    assert n["unexplained"] == 0, n
    # end to end (oracle's own intermediates): the final screens agree except around those pixels
    ours = smaa.run(color, preset, area, search)
    d = np.abs(ours["screen"].astype(np.int16) - fx["screen"].astype(np.int16)).max(-1)
    assert (d > 1).sum() <= 3 * (b["differing"] + n["differing"] + e["differing"]) + 8, int((d > 1).sum())
    assert (d > 1).sum() <= 0.3 * n_edge                 # ... and so are the screen pixels they reach (<= 24 % of the edge pixels)


def test_search_table_equals_the_reference_table_and_live_run_with_the_real_tables(built):
    """Build container only: tests/smaa_tables.search_table() == searchTexBytes of the reference; and the whole accounting again
    with the reference's REAL area table (read in place, never stored)."""
    ref_smaa = pytest.importorskip("oracle.ref_gl.ref_smaa")
    from oracle.ref_gl import ref_gl
    if not ref_gl.available():
        pytest.skip("needs /root/reference and Mesa llvmpipe (build container)")
    os.environ.setdefault("GALLIVM_PERF", "no_aos_sampling,no_quad_lod")
    area, search = ref_smaa.reference_luts()
    assert np.array_equal(smaa_tables.search_table(), search)
    assert np.array_equal(smaa_tables.area_table(), area)     # the library's generated area table IS the reference's (tests/test_smaa_tables.py)
    color = smaa_cases.pattern(11, 240, 150)
    for preset in ("ULTRA", "MEDIUM"):
        r = ref_smaa.run(color, preset, area, search)
        assert smaa_classify.classify_edges(color, r["edges"], preset)["unexplained"] == 0
        b = smaa_classify.classify_blend(r["edges"], r["blend"], preset, area, search)
        assert b["unexplained"] == 0, b
        assert smaa_classify.classify_neighborhood(color, r["blend"], r["screen"])["unexplained"] == 0


def test_reference_run_is_reproducible(built, tables):
    """llvmpipe today still produces the committed vectors (build container only)."""
    ref_smaa = pytest.importorskip("oracle.ref_gl.ref_smaa")
    from oracle.ref_gl import ref_gl
    if not ref_gl.available():
        pytest.skip("needs /root/reference and Mesa llvmpipe (build container)")
    os.environ.setdefault("GALLIVM_PERF", "no_aos_sampling,no_quad_lod")
    fx = np.load(os.path.join(GOLD, "smaa_ref_pattern_low.npz"))
    r = ref_smaa.run(fx["color"], "LOW", *tables)
    for k in ("edges", "blend", "screen"):
        assert np.array_equal(r[k], fx[k]), k


# ---- properties that need no reference ---------------------------------------------------------------------------------
def test_flat_and_sub_threshold_images_pass_through_unchanged(built, tables):
    rng = np.random.default_rng(5)
    img = np.full((40, 64, 4), 255, np.uint8)
    img[..., :3] = (90, 120, 60)
    img[..., :3] += rng.integers(0, 6, (40, 64, 3)).astype(np.uint8)      # luma steps of at most 5/255 < every threshold
    for preset in smaa.PRESETS:
        r = smaa.run(img, preset, *tables)
        assert not r["edges"].any() and not r["blend"].any() and np.array_equal(r["screen"], img)


def test_passes_commute_with_the_symmetries_the_shader_has(built, tables):
    """Pass 1 treats left/top alike: transposing the frame swaps the two edge channels (luma edge detection has no other
    orientation dependence); pass 3 with all-zero weights is the identity."""
    img = smaa_cases.pattern(21, 96, 96)
    a = smaa.run(img, "HIGH", *tables)["edges"]
    b = smaa.run(np.ascontiguousarray(img.transpose(1, 0, 2)), "HIGH", *tables)["edges"]
    assert np.array_equal(a, b.transpose(1, 0, 2)[..., ::-1])
    assert np.array_equal(smaa.neighborhood_pass(img, np.zeros_like(img)), img)


def test_weights_only_where_there_are_edges_and_alpha_passes_through(built, tables):
    img = smaa_cases.pattern(22, 160, 90)
    for preset in smaa.PRESETS:
        r = smaa.run(img, preset, *tables)
        has_w = r["blend"].any(-1)
        assert not (has_w & ~r["edges"].any(-1)).any()                    # exact pixel positions: no phantom edges
        assert (r["screen"][..., 3] == 255).all()                         # alpha 255 blended with alpha 255
        changed = (r["screen"] != img).any(-1)
        near = np.zeros_like(has_w)
        p = np.pad(has_w, 1)
        for dy in (0, 1, 2):
            for dx in (0, 1, 2):
                near |= p[dy:dy + has_w.shape[0], dx:dx + has_w.shape[1]]
        assert not (changed & ~near).any()                                # only pixels next to a weight are re-blended
