"""Pixel-by-pixel accounting of the differences between the tracer oracle and the REFERENCE's own shader run on Mesa llvmpipe
(tests/golden/ref_frame_*.npz). Replaces the "at most N % of the pixels may differ" limits of round 1: every pixel that is further
than 1e-4 from the reference has to be claimed by one of the mechanisms below, each of which is a place where the GLSL text leaves the
result to the implementation; a pixel claimed by none fails the test.

 unstable   The oracle's OWN answer at that pixel changes (by more than STABLE_TOL + GRADIENT x displacement) when every primary ray is displaced by a few float
            ulps (orc_set_ray_jitter, +-2^-14 .. 2^-10 of a pixel): a hit/miss, root-selection or branch decision of rt.frag sits within
            rounding of flipping. llvmpipe evaluates normalize() as v * rsqrt(dot), fuses and reorders; the discriminant of the far
            planets cancels catastrophically (rt.frag:345-347), Durand-Kerner stops at 1e-3 (rt.frag:479). Silhouettes, shadow edges and
            every torus pixel whose accepted root depends on the iteration's last sweep land here.
            BOUNDED (round 3): being unstable excuses nothing by itself -- the REFERENCE's pixel and the CANDIDATE's pixel must each be
            within NEAR_TOL = 2e-3 of ONE of the oracle's own 41 renders of that pixel (the undisplaced one and the 40 displaced ones),
            i.e. each must be one of the answers the shader text gives within rounding there -- not merely somewhere between two of them:
            a pixel that flips between an object and the floor does not excuse a value in between. (And inside their envelope [min, max]
            widened by ENVELOPE_PAD, which the first condition all but implies.) A defect inside a silhouette is not excused.
 torus      A Durand-Kerner root was accepted somewhere on the pixel's path: the root is only good to the solver's stop criterion
            (|delta| < 1e-3 in t, rt.frag:470-481), shading and secondary rays inherit that. Bounded: <= TORUS_TOL = 5e-3 in colour; the
            ROOTS themselves are pinned by test_accepted_torus_roots_equal_the_reference_shaders (the reference's shader instrumented to
            emit them: <= 2e-3, and nothing of this class is left once the reference's root is substituted).
 t21        The path re-entered a box at a negative distance (trap T21) -- with a refractive box this is the reference's degenerate
            geometric series: bounded, <= T21_TOL.
 divergent  A mip-mapped fetch for which a neighbour of the 2x2 pixel quad did not execute the same fetch (object and shadow boundaries,
            different bounce depths): GLSL leaves derivatives undefined in non-uniform control flow (GLSL 4.50 section 8.13.1). The oracle's
            rule takes that derivative as 0; llvmpipe differences whatever its masked-off lanes hold, i.e. it samples
            at SOME level of detail. BOUNDED (round 3): the reference's pixel and the candidate's pixel must lie inside the envelope of
            the oracle's renders with every mip-mapped fetch forced to level 0, 1, ..., top (orc_set_lod_force; a trilinear sample is
            piecewise linear in the level with knots at the integers, so these renders bracket every level), widened by LOD_PAD.
 box_nan    A NaN operand (0 * inf for a ray parallel to a box face, trap T5) entered intersectBox's min / max chains (rt.frag:412-413):
            GLSL leaves min / max of a NaN undefined -- the oracle follows the specification's wording, llvmpipe's SSE min/max return the
            second operand -- so whether that box is hit is the implementation's choice. No value bound; the tests demand that NO pixel of
            the committed fixtures needs this class (it exists for tools/fuzz_reference.py's degenerate scenes).
 quad_neighbour  A mip-mapped fetch whose 2x2 quad holds an unstable or divergent pixel: its derivatives difference that pixel's uv.
            BOUNDED like `divergent`: inside the forced-level envelope.
 divergent_alpha (round 5) A divergent / quad-neighbour pixel OUTSIDE that envelope whose path went through an alpha-textured ring
            (rt.frag:884: `alpha < 1` passes the ray through and weights what follows by 1 - alpha): the sampled alpha steers the PATH, so
            the level llvmpipe took for that ONE fetch and the levels of all other fetches enter the colour independently, and renders that
            force every fetch to the same level do not bracket it. BOUNDED: inside the envelope of the oracle's renders with the ring's
            hit-site fetch forced to level i, its shadow-site fetches (inShadow adds the ring's alpha, rt.frag:647) to level k and every other fetch
            to level j, for all triples (i, k, j) (orc_set_lod_force_site; only the quad
            rows of the pixels in question are rendered), widened by LOD_PAD. This is the class of round 4's one LEFTOVER pixel per
            full-size fixture.
 approx_math (last) A stable pixel within APPROX_TOL = 5e-4: llvmpipe's pow / exp / log2 are polynomial approximations (measured: pow 1e-5
            relative at small exponents; a specular pow(x, 200) amplifies that 200-fold). Counted and bounded.
 unstable_between (last resort, counted) unstable, inside the envelope of the 41 renders, but not within NEAR_TOL of any: <= 8 pixels per frame.
 edge       (last resort) The oracle's frame has a jump (> JUMP_TOL) between this pixel and a 4-neighbour and the reference's pixel equals the
            oracle's on the other side (<= EDGE_TOL): a silhouette or shadow boundary the two place less than one pixel apart without the
            oracle's own decision being within jitter range. Counted; the tests bound how many there may be.
 texture    A mip-mapped 2-D texture was sampled with all quad neighbours present: what is left is the GL implementation's atan/asin/log2
            and filter arithmetic. Bounded: <= tex_tol -- 0.01 with level 0 only on both sides, 0.1 with the same mip texels and llvmpipe's
            level formula on both sides (the *_same_mips fixtures, and the plain fixtures with llvmpipe's own generated levels read back
            into the oracle: gl_mips), 0.1 for the product's rule against the plain run.
 texture_level  (only where the caller says the two sides select levels differently) beyond tex_tol but inside the forced-level envelope.
Everything else ("strict") must be within 1e-4 -- float noise."""
from __future__ import annotations

import itertools

import numpy as np

from oracle import oracle

TOL = 1e-4
STABLE_TOL = 5e-5
GRADIENT = 0.5       # colour change per pixel that still counts as smooth shading: a displacement of p pixels may move the answer by GRADIENT * p
JITTER_PX = (2.0 ** -14, 2.0 ** -12, 2.0 ** -10)
JUMP_PX, JUMP_TOL = (2.0 ** -8, 2.0 ** -6), 0.02    # larger displacements count only if the answer JUMPS (a smooth gradient moves < 0.01 over 2^-6 px)
TORUS_TOL = 5e-3     # measured: 69 such pixels on the 64-torus frame, all <= 2.4e-4 but one at 3.4e-3 (round 2 allowed 0.05)
T21_TOL = 0.02
EDGE_TOL = 1e-3
TEX_LEVEL_PAD = 6e-3 # texture_level pixels: inside the forced-level envelope widened by one LSB of a mip texel (1/255) + LOD_PAD
LOD_PAD = 2e-3       # divergent / quad_neighbour pixels: reference and candidate within the forced-level envelope widened by this
NEAR_TOL = 2e-3      # an unstable pixel's reference AND candidate values must each be this close to one of the oracle's 41 renders of the pixel
ENVELOPE_PAD = 1e-4  # an unstable pixel's reference AND candidate values must lie within the oracle's jitter envelope widened by this
PAIR_MAX_PIXELS = 64 # divergent_alpha is evaluated only for a frame that sends at most this many pixels there: each quad row costs (top + 1)^3 two-row
                     # renders, the class exists for a handful of pixels per frame (the committed fixtures need 0 ... 8), and a frame that sends hundreds --
                     # the damaged frames of test_defects_inside_the_permissive_sets_are_not_excused -- keeps them unexplained: the verdict it should get
APPROX_TOL = 5e-4    # llvmpipe's pow / exp are polynomial approximations: pow(x, 200) of a specular term is good to ~1e-3 of its value


_SAMPLES = {}  # per probe: the 41 renders (41, H, W, 3) float32
_PROBES = {}   # per fixture and LOD mode: the oracle's frame, its event tags and its instability mask (shared by the oracle / host / GPU tests)


def _diff(a, b):
    d = np.abs(a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)).max(-1)
    return np.where(np.isnan(d), np.inf, d)


def probe(ref: dict, texture_lod: int = 1, gl_mips: bool = False, threads: int = 8):
    """The oracle's diagnostics for one fixture: (base frame, event tags, unstable mask, jitter envelope lo / hi, NaN seen, forced-level
    envelope lo / hi or None). gl_mips: the oracle samples the mip levels the GL implementation generated (fixture, plain textured runs)
    instead of its own integer means -- with texture_lod = 2 that is "llvmpipe's texels and llvmpipe's level formula"."""
    _build_probe(ref, texture_lod, gl_mips, threads)
    return _PROBES[(id(ref["frame"]), ref.get("name"), texture_lod, gl_mips)]


def _build_probe(ref, texture_lod, gl_mips, threads):
    w, h = ref["width"], ref["height"]
    key = (id(ref["frame"]), ref.get("name"), texture_lod, gl_mips)
    if key not in _PROBES:
        oracle.OracleScene.drop_mips()      # no override left behind by another probe
        O = oracle.OracleScene(ref["scene"], w, h, ref["textures"], ref["cubemap"], texture_lod=texture_lod, cube_mipmap=ref.get("cube_mipmap", False))
        if gl_mips:
            assert ref.get("gl_mips"), "this fixture holds no GL mip levels"
            for uniform, levels in ref["gl_mips"].items():
                O.set_mip_levels(uniform, levels)
        tags = np.zeros((h, w), np.uint32)
        base, _ = O.render(threads=threads, tags=tags)
        unstable = np.zeros((h, w), bool)
        lo = np.where(np.isnan(base[..., :3]), np.inf, base[..., :3]).astype(np.float64)
        hi = np.where(np.isnan(base[..., :3]), -np.inf, base[..., :3]).astype(np.float64)
        nan_seen = np.isnan(base[..., :3]).any(-1)
        samples = [base[..., :3].copy()]
        for px, tol in [(p, STABLE_TOL + GRADIENT * p) for p in JITTER_PX] + [(p, JUMP_TOL) for p in JUMP_PX]:
            dj = px / h
            for jx, jy in itertools.product((-dj, 0.0, dj), repeat=2):
                if jx == 0.0 and jy == 0.0:
                    continue
                ij, _ = O.render(threads=threads, jitter=(jx, jy))
                unstable |= _diff(ij, base) > tol
                v = ij[..., :3].astype(np.float64)
                samples.append(ij[..., :3].copy())
                lo, hi = np.fmin(lo, v), np.fmax(hi, v)        # (fmin / fmax skip NaN renders; those pixels are noted in nan_seen)
                nan_seen |= np.isnan(v).any(-1)
        lod_lo = lod_hi = None
        if texture_lod and (tags & oracle.TAG_TEXTURE).any():
            # every mip-mapped fetch at level 0, 1, ..., top: the bracket of whatever level an implementation took (divergent quads)
            top = max([int(np.ceil(np.log2(max(img.shape[0], img.shape[1])))) for _u, _n, img in ref["textures"]]
                      + ([int(np.ceil(np.log2(ref["cubemap"][0].shape[0])))] if ref.get("cube_mipmap") else []))
            lod_lo, lod_hi = lo.copy(), hi.copy()
            for level in range(top + 1):
                il, _ = O.render(threads=threads, lod_force=float(level))
                v = il[..., :3].astype(np.float64)
                lod_lo, lod_hi = np.fmin(lod_lo, v), np.fmax(lod_hi, v)
        if gl_mips:
            oracle.OracleScene.drop_mips()  # the override must not reach renders other tests make from the same texture arrays
        _SAMPLES[key] = np.stack(samples)
        _PROBES[key] = (base, tags, unstable, ref["frame"], lo, hi, nan_seen, lod_lo, lod_hi)     # (the frame is kept so that its id stays unique)


def _pair_envelope_ok(ref, texture_lod, gl_mips, threads, cand, img):
    """For the pixels of `cand`: are the reference's and the candidate's values inside the envelope of the oracle's renders with the ring's
    hit-site fetch at level i, its shadow-site fetches at level k and every other mip-mapped fetch at level j, over all (i, k, j)? Renders only the quad rows involved."""
    w, h = ref["width"], ref["height"]
    top = max(int(np.ceil(np.log2(max(im.shape[0], im.shape[1])))) for _u, _n, im in ref["textures"])
    oracle.OracleScene.drop_mips()
    O = oracle.OracleScene(ref["scene"], w, h, ref["textures"], ref["cubemap"], texture_lod=texture_lod, cube_mipmap=ref.get("cube_mipmap", False))
    if gl_mips:
        for uniform, levels in ref["gl_mips"].items():
            O.set_mip_levels(uniform, levels)
    ok = np.zeros((h, w), bool)
    width = np.zeros((h, w))      # how wide the envelope is at each pixel (largest channel): a claim inside a WIDE envelope says little, so it is reported
    try:
        for y0 in sorted({int(y) & ~1 for y in np.argwhere(cand)[:, 0]}):
            y1 = min(y0 + 2, h)
            lo = np.full((y1 - y0, w, 3), np.inf); hi = np.full((y1 - y0, w, 3), -np.inf)
            for i in range(top + 1):
                for k in range(top + 1):
                    for j in range(top + 1):
                        v = O.render(y0, y1, threads=threads, lod_force=float(j), lod_force_site=(("texture_ring", 0, float(i)), ("texture_ring", 1, float(k))))[0][..., :3].astype(np.float64)
                        lo, hi = np.fmin(lo, v), np.fmax(hi, v)
            r = ref["frame"][y0:y1, :, :3].astype(np.float64); c = img[y0:y1, :, :3].astype(np.float64)
            ok[y0:y1] = ((r >= lo - LOD_PAD) & (r <= hi + LOD_PAD)).all(-1) & ((c >= lo - LOD_PAD) & (c <= hi + LOD_PAD)).all(-1)
            width[y0:y1] = (hi - lo).max(-1)
    finally:
        if gl_mips:
            oracle.OracleScene.drop_mips()
    return ok, width


def classify(ref: dict, candidate: np.ndarray | None = None, texture_lod: int = 1, tex_tol: float = 0.0, threads: int = 8, gl_mips: bool = False,
             tex_level_envelope: bool = False) -> dict:
    """ref: tests/reference_frames.load(name). candidate: the frame under test (default: the oracle's own render).
    tex_tol: bound for pixels of the `texture` category (0 = none may differ). gl_mips: see probe(). tex_level_envelope: adds the class
    `texture_level` (see below) -- for comparisons in which the two sides are KNOWN to select levels differently (the product's rule
    against llvmpipe's own run). Returns counts per category and `unexplained` (must be 0) with up to 8 (x, y, difference, tags)."""
    w, h = ref["width"], ref["height"]
    base, tags, unstable, _, lo, hi, nan_seen, lod_lo, lod_hi = probe(ref, texture_lod, gl_mips, threads)
    img = base if candidate is None else candidate
    d = _diff(img, ref["frame"])
    bad = d > TOL
    out = dict(pixels=w * h, over=int(bad.sum()), max=float(d.max()))
    def in_envelope(frame):
        v = frame[..., :3].astype(np.float64)
        ok = ((v >= lo - ENVELOPE_PAD) & (v <= hi + ENVELOPE_PAD)).all(-1)
        return ok | (np.isnan(v).any(-1) & nan_seen)           # NaN is an answer only where one of the oracle's renders gave NaN
    S = _SAMPLES[(id(ref["frame"]), ref.get("name"), texture_lod, gl_mips)]

    def near_a_sample(frame):
        v = frame[..., :3].astype(np.float32)
        best = np.full((h, w), np.inf)
        for k in range(S.shape[0]):
            dd = np.abs(S[k] - v).max(-1)
            best = np.fmin(best, np.where(np.isnan(dd), np.inf, dd))
        return (best <= NEAR_TOL) | (np.isnan(v).any(-1) & nan_seen)
    hull = unstable & in_envelope(ref["frame"]) & in_envelope(img)
    excused = hull & near_a_sample(ref["frame"]) & near_a_sample(img)
    left = bad & ~excused
    out["unstable"] = int((bad & excused).sum())
    out["unstable_outside_envelope"] = int((bad & unstable & ~excused).sum())   # these go on to the bounded categories below
    out["unstable_pixels_in_frame"] = int(unstable.sum())

    def claim(name, mask):
        nonlocal left
        c = left & mask
        out[name] = int(c.sum())
        left = left & ~c
    def in_lod_envelope(frame):
        if lod_lo is None:
            return np.zeros((h, w), bool)
        v = frame[..., :3].astype(np.float64)
        return ((v >= lod_lo - LOD_PAD) & (v <= lod_hi + LOD_PAD)).all(-1)
    lod_ok = in_lod_envelope(ref["frame"]) & in_lod_envelope(img)
    claim("divergent", ((tags & oracle.TAG_QUAD_DIVERGENT) != 0) & lod_ok)
    claim("torus", ((tags & oracle.TAG_TORUS) != 0) & (d <= TORUS_TOL))
    claim("t21", ((tags & oracle.TAG_BOX_INSIDE) != 0) & (d <= T21_TOL))
    claim("box_nan", (tags & oracle.TAG_BOX_NAN) != 0)
    # a mip-mapped fetch differences the uv of its 2x2-quad neighbours: if one of those is itself unstable or divergent, so is this LOD
    flagged = unstable | ((tags & oracle.TAG_QUAD_DIVERGENT) != 0)
    hq, wq = (h // 2) * 2, (w // 2) * 2
    quad_any = np.zeros_like(flagged)
    q = flagged[:hq, :wq].reshape(hq // 2, 2, wq // 2, 2).any(axis=(1, 3))
    quad_any[:hq, :wq] = np.repeat(np.repeat(q, 2, axis=0), 2, axis=1)
    claim("quad_neighbour", ((tags & oracle.TAG_TEXTURE) != 0) & quad_any & lod_ok)
    # divergent_alpha: what the two classes above could not bracket, against the envelope over independent levels for the ring's alpha fetch
    # and for everything else -- rendered on demand, only the quad rows that hold such a pixel
    cand = left & ((tags & oracle.TAG_TEXTURE) != 0) & (((tags & oracle.TAG_QUAD_DIVERGENT) != 0) | quad_any)
    out["divergent_alpha"] = 0
    out["divergent_alpha_candidates"] = int(cand.sum())
    out["divergent_alpha_envelope_width"] = 0.0      # the widest envelope a claim of this class was made in (a wide one excuses a lot)
    if cand.any() and int(cand.sum()) <= PAIR_MAX_PIXELS and lod_lo is not None and any(u == "texture_ring" for u, _n, _i in ref["textures"]):
        pair_ok, pair_width = _pair_envelope_ok(ref, texture_lod, gl_mips, threads, cand, img)
        claimed = left & cand & pair_ok
        claim("divergent_alpha", cand & pair_ok)
        if claimed.any():
            out["divergent_alpha_envelope_width"] = float(pair_width[claimed].max())
    claim("texture", ((tags & oracle.TAG_TEXTURE) != 0) & (d <= tex_tol))
    if tex_level_envelope and lod_lo is not None:
        # texture_level: a mip-mapped fetch whose value differs by more than tex_tol although every quad neighbour was present: the two
        # sides chose different LEVELS (llvmpipe's piecewise-linear log2 and float mip averages against the product's rule, DESIGN.md
        # section 9) -- accepted only inside the forced-level envelope widened by TEX_LEVEL_PAD (one LSB of a mip texel + LOD_PAD)
        def in_wide(frame):
            v = frame[..., :3].astype(np.float64)
            return ((v >= lod_lo - TEX_LEVEL_PAD) & (v <= lod_hi + TEX_LEVEL_PAD)).all(-1)
        claim("texture_level", ((tags & oracle.TAG_TEXTURE) != 0) & in_wide(ref["frame"]) & in_wide(img))
    else:
        out["texture_level"] = 0
    # edge: the oracle has a discontinuity between this pixel and a 4-neighbour, and the reference's pixel equals the oracle's pixel on the
    # other side of it -- an edge (silhouette, shadow boundary) that the two place less than one pixel apart
    if left.any():
        B = base[..., :3].astype(np.float64)
        R = ref["frame"][..., :3].astype(np.float64)
        edge = np.zeros_like(left)
        for dy, dx in ((0, 1), (0, -1), (1, 0), (-1, 0)):
            N = np.roll(B, (dy, dx), axis=(0, 1))
            jump = np.abs(N - B).max(-1) > JUMP_TOL
            same = np.abs(N - R).max(-1) <= EDGE_TOL
            ok = jump & same
            if dy == 1: ok[0, :] = False
            if dy == -1: ok[-1, :] = False
            if dx == 1: ok[:, 0] = False
            if dx == -1: ok[:, -1] = False
            edge |= ok
        claim("edge", edge)
    else:
        out["edge"] = 0
    # unstable_between (last resort, counted): an unstable pixel inside the envelope of the 41 renders but further than NEAR_TOL from each --
    # the answer wanders continuously there (a torus root's last sweep feeding a mirror ray) and 41 samples do not pin it; the tests bound
    # how many such pixels a frame may have (measured: 4 on the 64-torus frame, 0 elsewhere)
    claim("unstable_between", hull)
    claim("approx_math", d <= APPROX_TOL)
    out["unexplained"] = int(left.sum())
    out["where"] = [(int(x), int(y), float(d[y, x]), int(tags[y, x])) for y, x in np.argwhere(left)[:8]]
    return out
