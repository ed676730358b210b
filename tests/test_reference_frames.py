"""Parity against the REFERENCE ITSELF: tests/golden/ref_frame_*.npz hold the output of the reference's own fragment
shader (assets/shaders/rt.frag, executed on Mesa llvmpipe by oracle/ref_gl + tools/gen_reference_frames.py in the build
container). The oracle, the host build of the product's device code and -- on the GPU box -- the HIP kernel are compared
with those pixels PIXEL BY PIXEL: every pixel further than 1e-4 from the reference must be claimed by one of the
implementation-defined mechanisms of tests/reference_classify.py (decision within rounding of flipping, Durand-Kerner's 1e-3 stop,
the degenerate T21 series, NaN through min/max, derivatives in divergent quads, the sampler's level selection, polynomial pow/exp);
a pixel claimed by none fails. The two textured scenes are pinned three ways (tests/reference_frames.py VARIANTS): with level 0
only on both sides (everything but the mip machinery: strict), with the oracle's mip texels handed to GL and the oracle in
llvmpipe's LOD mode, and as the reference really runs."""
import os

import numpy as np
import pytest

import harness
import reference_classify as rc
import reference_frames as rf
from oracle import oracle

NAMES = [n for n in rf.CASES if n not in rf.FULL_SIZE]
UNTEXTURED = [n for n in NAMES if not rf.CASES[n][1]]
# (fixture, oracle texture_lod, bound for the `texture` category, oracle samples llvmpipe's generated mip levels?, texture_level class?)
#   *_level0     level 0 only on both sides: everything but the mip machinery, 0.01
#   *_same_mips  GL was handed the oracle's mip texels, oracle in llvmpipe's level formula: 0.1
#   plain, "gl"  the reference as it really runs (glGenerateMipmap) against the oracle with llvmpipe's generated levels read back + llvmpipe's
#                level formula: the same bound, 0.1 (round 2 had 1.0 here, i.e. none)
#   plain, rule  ... against the oracle's / the product's own texture rule (integer-mean mips, exact log2): 0.1, or a sample of the same
#                texture at another level (texture_level)
TEXTURED_PLAN = ([(c + "_level0", 0, 0.01, False, False) for c in rf.TEXTURED] + [(c + "_same_mips", 2, 0.1, False, False) for c in rf.TEXTURED]
                 + [(c, 2, 0.1, True, True) for c in rf.TEXTURED + rf.TEXTURED_PLAIN_ONLY] + [(c, 1, 0.1, False, True) for c in rf.TEXTURED + rf.TEXTURED_PLAIN_ONLY])
PLAN_IDS = [p[0] + ("_gl_mips" if p[3] else "") + ("_product_rule" if p[4] else "") for p in TEXTURED_PLAN]


# The two full-size fixtures (round 4: 307 200 and 921 600 pixels against 36 864 ... 76 800 of the others) each left ONE pixel that no class
# claimed and, in the 1280 x 720 frame, one pixel of the box_nan class. Round 5:
#   config0_full      (281, 151), 0.12 off: a mip-mapped ring fetch in a divergent quad whose sampled alpha decides the `alpha < 1` pass-through
#                     (rt.frag:884), so the pixel's PATH depends on the level llvmpipe took for that one fetch. Claimed now by the class
#                     divergent_alpha (tests/reference_classify.py: levels of the ring's hit-site fetch, of its shadow-site fetches and of
#                     everything else forced independently) -- nothing is left over in this frame;
#   app_default_full  (144, 382), 0.011 off, a torus pixel (TORUS_TOL is 5e-3): the accepted root's 1e-3 reaches the colour through a mirror
#                     bounce. Kept as the one named exception, with its mechanism TESTED: test_the_leftover_torus_pixel_is_the_solvers_tolerance
#                     moves the camera ray's root by the solver's own tolerance and the reference's colour lies between the oracle's answers;
#                     and one pixel where a ray runs parallel to a face of the glass box (0 * inf in intersectBox, trap T5: min / max of a NaN).
LEFTOVER = {"app_default_full": dict(unexplained=1, box_nan=1)}


def _accept(name, r, textured):
    px = r["pixels"]
    allow = LEFTOVER.get(name, dict(unexplained=0, box_nan=0))
    assert r["unexplained"] <= allow["unexplained"], f"{name}: {r['unexplained']} pixels differ from the reference shader by more than 1e-4 and no mechanism claims them: {r['where']}"
    assert r["edge"] <= 3 and r["approx_math"] <= max(3, px // 2000) and r["unstable_between"] <= 8, (name, r)   # the last-resort categories stay marginal
    assert r["unstable_pixels_in_frame"] <= 0.12 * px, (name, r)                         # the envelope-bounded set is a small part of the frame
    assert r["box_nan"] <= allow["box_nan"], (name, r)                                   # the one class without a value bound is not needed by any quarter-size fixture
    #  the ring-alpha class accepts a pixel anywhere inside an envelope over (top + 1)^3 forced-level renders, which is wide where the
    # alpha steers the path: it stays a handful of pixels in EVERY fixture, and how wide the widest envelope of a claim was is part of the record
    assert r.get("divergent_alpha", 0) <= 8, (name, r)
    if r.get("divergent_alpha", 0):
        print(f"{name}: {r['divergent_alpha']} pixel(s) claimed by divergent_alpha, widest envelope {r['divergent_alpha_envelope_width']:.3f}")
    if not textured:
        assert r["divergent"] == 0 and r["texture"] == 0 and r["quad_neighbour"] == 0 and r["texture_level"] == 0, (name, r)
    return r


@pytest.mark.parametrize("name", UNTEXTURED)
def test_oracle_matches_reference_shader(built, name):
    ref = rf.load(name)
    assert "llvmpipe" in ref["renderer"]
    _accept(name, rc.classify(ref), False)


@pytest.mark.parametrize("name,lod,tex_tol,gl_mips,level_env", TEXTURED_PLAN, ids=PLAN_IDS)
def test_oracle_matches_reference_shader_textured(built, name, lod, tex_tol, gl_mips, level_env):
    r = _accept(name, rc.classify(rf.load(name), texture_lod=lod, tex_tol=tex_tol, gl_mips=gl_mips, tex_level_envelope=level_env), True)
    if gl_mips:                        # llvmpipe's texels + llvmpipe's level formula: the plain run is as close as the *_same_mips one;
        # texture_level then only for the odd pixel on a planet's u = 0 / 1 seam, where rt.frag:327's `df.x > 0.5` test is decided by WHICH
        # row / column of the quad an implementation differences (GLSL 4.50 section 8.13.1 allows either): config0 has one such pixel
        assert r["over"] <= 0.06 * r["pixels"] and r["texture_level"] <= 3, (name, r)
    elif level_env:
        assert r["texture_level"] <= 0.001 * r["pixels"], (name, r)
    if name.endswith("_level0"):       # without mip maps the textured frames are as close as the untextured ones
        assert r["over"] <= 0.012 * r["pixels"] and r["divergent"] == 0, (name, r)
    if name.endswith("_same_mips"):    # same texels + llvmpipe's LOD formula: well under the plain run
        plain = rc._diff(oracle.OracleScene(*_scene_args(rf.load(name[:-10])), texture_lod=1).render(threads=8)[0], rf.load(name[:-10])["frame"])
        assert r["over"] < 0.8 * int((plain > rc.TOL).sum()), (name, r)


# Round 5: the two textured plain runs again with BAND-LIMITED textures (no per-texel grain, no steps:
# textures.default_texture_set(smooth=True)). Which mip level an implementation takes then hardly matters, so the `texture` class is held
# at 5e-3 -- twenty times tighter than the grainy fixtures' 0.1, which was mostly the textures' own 0.03 of per-texel noise seen through a
# different level. (plan: fixture, oracle LOD mode, llvmpipe's own mip levels read back?)
SMOOTH_TEX_TOL = 5e-3
SMOOTH_PLAN = [(n, 2, True) for n in rf.SMOOTH]      # (the product's own rule against these runs: GPU_PLAN below, on the HIP kernel's frame)


@pytest.mark.parametrize("name,lod,gl_mips", SMOOTH_PLAN, ids=[p[0] + ("_gl_mips" if p[2] else "_product_rule") for p in SMOOTH_PLAN])
def test_oracle_matches_reference_shader_with_band_limited_textures(built, name, lod, gl_mips):
    r = _accept(name, rc.classify(rf.load(name), texture_lod=lod, tex_tol=SMOOTH_TEX_TOL, gl_mips=gl_mips, tex_level_envelope=True), True)
    assert r["texture"] > 100, (name, r)                       # (there are mip-mapped pixels to judge)
    if gl_mips:     # llvmpipe's texels and llvmpipe's level formula: all but a handful of the mip-mapped pixels within 5e-3 (measured: 122 of 122
                    # and 561 of 568; the rest up to 1e-2 -- seen through the glass sphere -- and one pixel on a planet's u = 0 / 1 seam)
        assert r["texture_level"] <= 8, (name, r)
    else:           # the product's own rule (integer-mean mips, exact log2) against the plain run: a few more choose another level
        assert r["texture_level"] <= 0.001 * r["pixels"], (name, r)
    assert r["divergent_alpha"] <= 8, (name, r)                # the ring-alpha class stays a handful of pixels


# Round 5: the sky box loaded the way GLWrapper::load_cubemap(faces, genMipmap = true) loads it (GLWrapper.cpp:307-310) -- cube mips and a
# trilinear sky fetch (rt.frag:893). Objects untextured: the sky is the only mip-mapped fetch. Judged like the band-limited 2-D fixtures:
# pixels whose quad executes the sky fetch together within 5e-3 (the rule: quotient-rule derivatives of the face coordinates, DESIGN.md
# section 9), pixels of divergent quads inside the forced-level envelope. lod 2 = llvmpipe's level formula, lod 1 = the product's exact log2.
@pytest.mark.parametrize("name,lod", [(n, l) for n in rf.CUBE_MIPS for l in (2, 1)], ids=[f"{n}_lod{l}" for n in rf.CUBE_MIPS for l in (2, 1)])
def test_oracle_matches_reference_shader_with_cube_mips(built, name, lod):
    ref = rf.load(name)
    r = _accept(name, rc.classify(ref, texture_lod=lod, tex_tol=SMOOTH_TEX_TOL, tex_level_envelope=True), True)
    assert r["texture"] > 100, (name, r)                       # (there are mip-mapped sky pixels to judge)
    assert r["texture_level"] <= 0.001 * r["pixels"], (name, r)
    # and the mips matter in these frames: the oracle WITHOUT cube mips is further from the reference's pixels than the class bound on many pixels
    flat = oracle.OracleScene(*_scene_args(ref), texture_lod=1, cube_mipmap=False).render(threads=8)[0]
    assert int((rc._diff(flat, ref["frame"]) > SMOOTH_TEX_TOL).sum()) > 4 * int((rc._diff(rc.probe(ref, lod)[0], ref["frame"]) > SMOOTH_TEX_TOL).sum()), name


def _scene_args(ref):
    return ref["scene"], ref["width"], ref["height"], ref["textures"], ref["cubemap"]


def test_exact_cases_are_exact(built):
    """Where nothing implementation-defined is involved (no mip-mapped textures, no iterative solver, no degenerate glass
    box), the oracle reproduces the reference shader's pixels to float noise."""
    for name in ("trap_inside_box", "trap_degenerate_rings_untextured"):
        ref = rf.load(name)
        img, _ = oracle.OracleScene(*_scene_args(ref), texture_lod=1).render(0, ref["height"], threads=8)
        _f4, _f2, mx = rf.compare(img, ref["frame"])
        assert mx < 1e-4, (name, mx)


def test_the_classifier_does_not_excuse_a_wrong_frame(built):
    """The accounting must have teeth: damage the candidate in ways a real defect would (a colour channel scaled by 1 %, a primitive's
    pixels shifted by one column, shadows 5 % too dark over a stable region) and the same classifier must leave pixels unexplained."""
    ref = rf.load("default_untextured")
    img, _ = oracle.OracleScene(*_scene_args(ref)).render(threads=8)
    assert rc.classify(ref, candidate=img)["unexplained"] == 0
    for damage in ("gain", "shift", "region"):
        bad = img.copy()
        if damage == "gain":
            bad[..., 1] *= np.float32(1.01)
        elif damage == "shift":
            bad[40:90, 100:160] = img[40:90, 99:159]
        else:
            bad[20:60, 30:90, :3] *= np.float32(0.95)
        r = rc.classify(ref, candidate=bad)
        assert r["unexplained"] > 50, (damage, r)


def test_defects_inside_the_permissive_sets_are_not_excused(built):
    """Round 2's classifier excused ANY value on an unstable pixel and any value up to 1.0 on a mip-mapped one. Now: a defect confined to
    the unstable set of the torus frame (silhouettes, Durand-Kerner's last sweep) -- 3 % too dark there and nowhere else --, a defect
    confined to the mip-mapped pixels of the textured frame (an offset of 0.15; and of 0.02 in the level-0 variant), and a defect
    confined to the divergent-quad pixels (a value outside every level's sample) must each leave most of the damaged pixels
    unexplained."""
    ref = rf.load("torus")
    img, _ = oracle.OracleScene(*_scene_args(ref)).render(threads=8)
    _b, _t, unstable, *_ = rc.probe(ref, 1, False)
    assert unstable.sum() > 1000
    bad = img.copy()
    bad[unstable, :3] *= np.float32(0.97)
    r = rc.classify(ref, candidate=bad)
    lit = unstable & (img[..., :3].max(-1) > 0.25)           # 3 % of these is more than every bound (NEAR_TOL 2e-3, TORUS_TOL 5e-3)
    # (unexplained outright, or only "between" two of the oracle's answers -- a class the acceptance caps at 8 pixels per frame)
    assert lit.sum() > 1000 and r["unexplained"] + r["unstable_between"] > 0.9 * lit.sum() and r["unexplained"] > 1000, (r, int(lit.sum()))
    with pytest.raises(AssertionError):
        _accept("torus, damaged", r, False)
    for name, lod, off, kw in (("default", 1, 0.15, dict(tex_tol=0.1, tex_level_envelope=True)), ("default", 2, 0.15, dict(tex_tol=0.1, gl_mips=True)),
                               ("default_level0", 0, 0.02, dict(tex_tol=0.01))):
        ref = rf.load(name)
        base, tags, unstable, *_ = rc.probe(ref, lod, kw.get("gl_mips", False))
        tex = ((tags & oracle.TAG_TEXTURE) != 0) & ((tags & oracle.TAG_QUAD_DIVERGENT) == 0) & ~unstable
        assert tex.sum() > 1500
        bad = base.copy()
        bad[tex, :3] += np.float32(off)
        r = rc.classify(ref, candidate=bad, texture_lod=lod, **kw)
        assert r["unexplained"] > 0.8 * tex.sum(), (name, lod, r, int(tex.sum()))
    ref = rf.load("default")
    base, tags, unstable, *_ = rc.probe(ref, 1, False)
    div = ((tags & oracle.TAG_QUAD_DIVERGENT) != 0) & ~unstable
    assert div.sum() > 200
    bad = base.copy()
    bad[div, :3] = np.float32(1.5)                            # brighter than any texel of any level
    r = rc.classify(ref, candidate=bad, texture_lod=1, tex_tol=0.1, tex_level_envelope=True)
    assert r["unexplained"] > 0.9 * div.sum(), (r, int(div.sum()))


def test_accepted_torus_roots_equal_the_reference_shaders(built):
    """The `torus` class compares colours (<= 0.05). This compares ROOTS: the fixtures hold what the first calcInter of every pixel
    returned in the reference's own shader (instrumented at run time, oracle/ref_gl.instrument_primary_hit: t, type, num). (1) both sides
    hit the same primitive at every pixel that is not within rounding of a flip; (2) Durand-Kerner's accepted root agrees to 2e-3 (its
    stop criterion is max |delta| < 1e-3 over the four roots, rt.frag:479; near a double root -- a grazing ray -- the iteration converges
    linearly and stops further out) on 99.8 % of the stable pixels, to 2e-2 on all of them and to 1e-3 relative everywhere; the
    closed-form intersectors agree to 1e-4 relative; (3) with the reference's root SUBSTITUTED into the oracle, every stable pixel whose camera ray hits a torus is within
    1e-4 of the reference's colour -- nothing of the `torus` class is left, i.e. its <= 0.05 was all the solver's 1e-3."""
    for name in rf.PRIMARY_HITS:
        ref = rf.load(name)
        P = ref["primary"]
        O = oracle.OracleScene(*_scene_args(ref))
        frame, hits = O.primary_hits(threads=8)
        t, ty, nu = hits[..., 0].astype(np.float64), hits[..., 1].astype(np.int32), hits[..., 2].astype(np.int32)
        _b, _tags, unstable, *_ = rc.probe(ref, 1, False)
        same = (ty == P["type"]) & (nu == P["num"])
        assert not (~same & ~unstable).any(), (name, int((~same & ~unstable).sum()))
        assert (~same).sum() <= 8, (name, int((~same).sum()))
        tor = same & (ty == 4)
        assert tor.sum() > 500
        dt = np.abs(t - P["t"])
        stable = dt[tor & ~unstable]
        assert (stable > 2e-3).sum() <= 0.002 * stable.size and stable.max() <= 2e-2, (name, int((stable > 2e-3).sum()), float(stable.max()))
        assert (dt[tor] / np.maximum(1.0, np.abs(P["t"][tor]))).max() <= 1e-3
        other = same & (ty >= 0) & (ty != 4)
        assert (dt[other] / np.maximum(1.0, np.abs(P["t"][other]))).max() <= 1e-4, name
        sub, hits2 = O.primary_hits(threads=8, torus_t=np.where(P["type"] == 4, P["t"], 0.0).astype(np.float32))
        assert np.array_equal(hits2[..., 0][tor], P["t"][tor])
        m = tor & ~unstable
        before, after = rc._diff(frame, ref["frame"])[m], rc._diff(sub, ref["frame"])[m]
        assert after.max() <= 1e-4, (name, float(after.max()))
        if name == "torus":
            assert (before > 1e-4).sum() > 20       # (there WAS something to explain)


def test_the_leftover_torus_pixel_is_the_solvers_tolerance(built):
    """app_default_full (144, 382) is 0.011 from the reference, above the torus class' 5e-3, and no other class claims it (LEFTOVER). Its camera
    ray hits the torus, whose mirror term (reflect 0.2) carries a steep gradient there. Durand-Kerner stops when the roots move by less than
    1e-3 (rt.frag:475-481), so the accepted root is good to about that: with the oracle's own root displaced by -2e-3 ... +2e-3 (substituted
    through orc_set_primary_buffers, the mechanism of test_accepted_torus_roots_equal_the_reference_shaders) the pixel's colour sweeps a range
    that holds the reference's value -- i.e. the difference is the solver's tolerance and nothing else."""
    ref = rf.load("app_default_full")
    x, y = 144, 382
    O = oracle.OracleScene(*_scene_args(ref), texture_lod=1)
    y0 = y & ~1
    frame, hits = O.primary_hits(threads=2, y0=y0, y1=y0 + 2)
    t0, ty = float(hits[y, x, 0]), int(hits[y, x, 1])
    assert ty == 4, "the camera ray of the leftover pixel hits the torus"
    base = frame[y - y0, x, :3].astype(np.float64)
    want = ref["frame"][y, x].astype(np.float64)
    assert 5e-3 < np.abs(base - want).max() < 2e-2
    lo, hi = base.copy(), base.copy()
    for dt in (-2e-3, -1e-3, -5e-4, 5e-4, 1e-3, 2e-3):
        sub = np.zeros((ref["height"], ref["width"]), np.float32)
        sub[y, x] = np.float32(t0 + dt)
        f, _ = O.primary_hits(threads=2, torus_t=sub, y0=y0, y1=y0 + 2)
        v = f[y - y0, x, :3].astype(np.float64)
        lo, hi = np.minimum(lo, v), np.maximum(hi, v)
    assert ((want >= lo - 1e-4) & (want <= hi + 1e-4)).all(), (lo, hi, want)
    assert (hi - lo).max() > 5e-3          # the pixel really is that sensitive to the root


@pytest.mark.parametrize("name", rf.FULL_SIZE)
def test_oracle_is_within_the_limits_of_the_full_size_reference_frames(built, name):
    """The two configurations the reference is run at, at their own size and as it runs them (640 x 480 depth 1; 1280 x 720 depth 5 at t = 3;
    textured): here the oracle is held to the fixture's limits (fractions of pixels beyond 1e-4 / 1e-2, the same as the quarter-size
    'default' / 'config0' cases); the pixel-by-pixel accounting of these two -- for the HIP kernel's frame -- runs on the GPU box
    (test_hip_kernel_matches_reference_shader)."""
    ref = rf.load(name)
    img, _ = oracle.OracleScene(*_scene_args(ref), texture_lod=1).render(0, ref["height"], threads=os.cpu_count() or 1)
    f4, f2, _mx = rf.compare(img, ref["frame"])
    assert f4 <= ref["limits"][0] and f2 <= ref["limits"][1], (name, f4, f2)


@pytest.mark.parametrize("name", UNTEXTURED)
def test_product_device_code_on_host_matches_reference_shader(built, name):
    """The product's device header compiled for the host (no quads there: untextured cases only)."""
    ref = rf.load(name)
    img, _ = harness.render(*_scene_args(ref), cull=True)
    _accept(name, rc.classify(ref, candidate=img), False)


def test_reference_run_is_reproducible(built):
    """Build container only: executing the reference's shader again gives the committed pixels bit for bit."""
    from oracle.ref_gl import ref_gl
    if not ref_gl.available():
        pytest.skip("needs /root/reference and Mesa llvmpipe (build container only)")
    os.environ.setdefault("GALLIVM_PERF", "no_aos_sampling,no_quad_lod")
    name = "trap_inside_box"
    ref = rf.load(name)
    again, _ = ref_gl.render(ref["scene"], ref["width"], ref["height"], ref["textures"], ref["cubemap"])
    assert np.array_equal(again[..., :3], ref["frame"])


# the kernel has one texture rule (the product's: lod 1) and its level-0 mode: the level-0 fixtures at 0.01, the *_same_mips fixtures (GL had
# the very mip texels the kernel builds; only the level formula differs) and the plain runs at 0.1 or a sample of another level
GPU_PLAN = ([(n, 1, 0.0, False) for n in UNTEXTURED] + [(c + "_level0", 0, 0.01, False) for c in rf.TEXTURED]
            + [(c + "_same_mips", 1, 0.1, True) for c in rf.TEXTURED] + [(c, 1, 0.1, True) for c in rf.TEXTURED + rf.TEXTURED_PLAIN_ONLY + ("config0_full",)]
            + [(c, 1, SMOOTH_TEX_TOL, True) for c in rf.SMOOTH + rf.CUBE_MIPS])
# (app_default_full -- 1280x720, depth 5, the app's own pose -- is not in this plan since round 5: its accounting needs the oracle's 41 + 12 renders of
# that frame, 215 s of the GPU box's host CPU for one test, a third of the suite. The frame is pinned in two steps instead: the oracle against the
# reference's pixels by the full accounting in the CPU suite (test_oracle_is_within_the_limits_of_the_full_size_reference_frames), the HIP kernel
# against the oracle at the north star's 1e-4, pixel by pixel, with equal ray counts, in the test below.)


@pytest.mark.gpu
@pytest.mark.parametrize("name,lod,tex_tol,level_env", GPU_PLAN, ids=[p[0] for p in GPU_PLAN])
def test_hip_kernel_matches_reference_shader(built, name, lod, tex_tol, level_env):
    """The HIP kernel's frame through the same pixel-by-pixel accounting (the oracle only supplies the per-pixel event tags and the
    stability probe; the pixels judged are the GPU's)."""
    from raytracing_opengl_amd import wrapper
    ref = rf.load(name)
    gl = wrapper.make_renderer(*_scene_args(ref), texture_lod=1 if lod else 0, cube_mipmap=ref["cube_mipmap"])
    gl.draw()
    img = gl.read_pixels(wrapper.RTX_RGBA32F)
    gl.stop()
    _accept(name, rc.classify(ref, candidate=img, texture_lod=lod, tex_tol=tex_tol, tex_level_envelope=level_env), name not in UNTEXTURED)


@pytest.mark.gpu
def test_hip_kernel_equals_the_oracle_on_the_apps_full_size_frame(built):
    """app_default_full on the GPU: the HIP frame against the oracle's frame of the same inputs, max |difference| <= 1e-4, NaNs in the same places,
    ray counts equal (what the full accounting of this fixture rests on is checked in the CPU suite, see GPU_PLAN)."""
    from raytracing_opengl_amd import wrapper
    ref = rf.load("app_default_full")
    gl = wrapper.make_renderer(*_scene_args(ref), texture_lod=1)
    gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    gl.draw()
    img = gl.read_pixels(wrapper.RTX_RGBA32F)
    st = gl.stats()
    gl.stop()
    want, cnt = oracle.OracleScene(*_scene_args(ref), texture_lod=1).render(threads=8)
    assert int((np.isnan(img) != np.isnan(want)).sum()) == 0
    d = np.abs(img - want)
    d = np.where(np.isnan(d), 0.0, d)
    assert float(d.max()) <= 1e-4, (float(d.max()), int((d > 1e-4).sum()))
    assert st["rays_closest"] == cnt["rays_closest"] and st["rays_shadow"] == cnt["rays_shadow"]


def test_default_scene_with_the_reference_asset_files(built):
    """Build container only, end to end on the reference's own inputs: its five texture files and six sky-box faces
    (JPEG + PNG), decoded by the shim's readers (include/rtx/jpeg_decode.h, png_decode.h -- byte-identical to the
    reference's stb_image, tests/test_jpeg_decode.py), go to (a) the reference's fragment shader on llvmpipe and (b) the
    oracle, judged pixel by pixel like the committed textured fixtures: a mip-mapped pixel within 0.1 or inside the envelope of the same
    texture's samples at the neighbouring levels (round 3 had 1.0 here, i.e. no bound): what differs is llvmpipe's mip rounding / LOD /
    atan on textured pixels and the silhouettes."""
    import ctypes
    from oracle import oracle
    from oracle.ref_gl import ref_gl
    from raytracing_opengl_amd import scenes, textures
    tex_dir = "/root/reference/assets/textures"
    if not ref_gl.available() or not os.path.isdir(tex_dir):
        pytest.skip("needs /root/reference and Mesa llvmpipe (build container only)")
    os.environ.setdefault("GALLIVM_PERF", "no_aos_sampling,no_quad_lod")
    lib = scenes._host_lib()
    lib.rtxh_decode_image.restype = ctypes.c_size_t
    lib.rtxh_decode_image.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                      ctypes.c_void_p, ctypes.c_size_t]

    def decode(path):
        w, h, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        need = lib.rtxh_decode_image(path.encode(), ctypes.byref(w), ctypes.byref(h), ctypes.byref(c), None, 0)
        assert need, path
        out = np.empty(need, np.uint8)
        lib.rtxh_decode_image(path.encode(), ctypes.byref(w), ctypes.byref(h), ctypes.byref(c), out.ctypes.data, need)
        return out.reshape(h.value, w.value, c.value)
    tex = []
    for (name, uniform, unit, w, h, c) in textures.REFERENCE_TEXTURES:      # main.cpp:149-153
        img = decode(os.path.join(tex_dir, name))
        assert img.shape == (h, w, c), (name, img.shape)
        tex.append((uniform, unit, img))
    faces = [decode(os.path.join(tex_dir, "sb_nebula", f"GalaxyTex_{s}{a}.jpg")) for a in "XYZ" for s in ("Positive", "Negative")]  # main.cpp:137-145
    assert all(f.shape == (textures.CUBEMAP_FACE, textures.CUBEMAP_FACE, 3) for f in faces)
    w, h = rf.W, rf.H
    sc = scenes.build_scene("default", w, h, 4)
    ref, missing = ref_gl.render(sc, w, h, tex, faces)
    assert not missing
    img, _ = oracle.OracleScene(sc, w, h, tex, faces, texture_lod=1).render()
    r = rc.classify(dict(scene=sc, width=w, height=h, textures=tex, cubemap=faces, frame=np.ascontiguousarray(ref[..., :3])), candidate=img, tex_tol=0.1, tex_level_envelope=True)
    # one shadow-edge pixel of the committed fixtures (162, 60: the two place a box's shadow boundary a pixel apart) is claimed there by
    # the `edge` rule; on the textured floor of this run its neighbour differs by the sampler's level selection too, so it is left over
    assert r["unexplained"] <= 2, (r["where"], {k: v for k, v in r.items() if k != "where"})
