"""Pixel-by-pixel accounting of the differences between the tracer oracle and the REFERENCE's own shader run on Mesa llvmpipe
(tests/golden/ref_frame_*.npz). Replaces the "at most N % of the pixels may differ" limits of round 1: every pixel that is further
than 1e-4 from the reference has to be claimed by one of the mechanisms below, each of which is a place where the GLSL text leaves the
result to the implementation; a pixel claimed by none fails the test.

 unstable   The oracle's OWN answer at that pixel changes (by more than STABLE_TOL + GRADIENT x displacement) when every primary ray is displaced by a few float
            ulps (orc_set_ray_jitter, +-2^-14 .. 2^-10 of a pixel): a hit/miss, root-selection or branch decision of rt.frag sits within
            rounding of flipping. llvmpipe evaluates normalize() as v * rsqrt(dot), fuses and reorders; the discriminant of the far
            planets cancels catastrophically (rt.frag:345-347), Durand-Kerner stops at 1e-3 (rt.frag:479). Silhouettes, shadow edges and
            every torus pixel whose accepted root depends on the iteration's last sweep land here.
 torus      A Durand-Kerner root was accepted somewhere on the pixel's path: the root is only good to the solver's stop criterion
            (|delta| < 1e-3 in t, rt.frag:470-481), shading and secondary rays inherit that. Bounded: <= TORUS_TOL.
 t21        The path re-entered a box at a negative distance (trap T21) -- with a refractive box this is the reference's degenerate
            geometric series: bounded, <= T21_TOL.
 divergent  A mip-mapped fetch for which a neighbour of the 2x2 pixel quad did not execute the same fetch (object and shadow boundaries,
            different bounce depths): GLSL leaves derivatives undefined in non-uniform control flow (GLSL 4.50 section 8.13.1). The oracle's
            rule takes that derivative as 0; llvmpipe differences whatever its masked-off lanes hold. Unbounded.
 box_nan    A NaN operand (0 * inf for a ray parallel to a box face, trap T5) entered intersectBox's min / max chains (rt.frag:412-413):
            GLSL leaves min / max of a NaN undefined -- the oracle follows the specification's wording, llvmpipe's SSE min/max return the
            second operand -- so whether that box is hit is the implementation's choice. Unbounded.
 quad_neighbour  A mip-mapped fetch whose 2x2 quad holds an unstable or divergent pixel: its derivatives difference that pixel's uv.
 approx_math (last) A stable pixel within APPROX_TOL = 5e-4: llvmpipe's pow / exp / log2 are polynomial approximations (measured: pow 1e-5
            relative at small exponents; a specular pow(x, 200) amplifies that 200-fold). Counted and bounded.
 edge       (last resort) The oracle's frame has a jump (> JUMP_TOL) between this pixel and a 4-neighbour and the reference's pixel equals the
            oracle's on the other side (<= EDGE_TOL): a silhouette or shadow boundary the two place less than one pixel apart without the
            oracle's own decision being within jitter range. Counted; the tests bound how many there may be.
 texture    A mip-mapped 2-D texture was sampled with all quad neighbours present: what is left is the GL implementation's atan/asin/log2
            and filter arithmetic. Bounded: <= tex_tol (caller's choice per fixture kind, see tests/test_reference_frames.py).
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
TORUS_TOL = 0.05
T21_TOL = 0.02
EDGE_TOL = 1e-3
APPROX_TOL = 5e-4    # llvmpipe's pow / exp are polynomial approximations: pow(x, 200) of a specular term is good to ~1e-3 of its value


_PROBES = {}   # per fixture and LOD mode: the oracle's frame, its event tags and its instability mask (shared by the oracle / host / GPU tests)


def _diff(a, b):
    d = np.abs(a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)).max(-1)
    return np.where(np.isnan(d), np.inf, d)


def classify(ref: dict, candidate: np.ndarray | None = None, texture_lod: int = 1, tex_tol: float = 0.0, threads: int = 8) -> dict:
    """ref: tests/reference_frames.load(name). candidate: the frame under test (default: the oracle's own render).
    tex_tol: bound for pixels of the `texture` category (0 = none may differ). Returns counts per category and `unexplained`
    (must be 0) with up to 8 (x, y, difference, tags)."""
    w, h = ref["width"], ref["height"]
    key = (id(ref["frame"]), ref.get("name"), texture_lod)
    if key not in _PROBES:
        O = oracle.OracleScene(ref["scene"], w, h, ref["textures"], ref["cubemap"], texture_lod=texture_lod)
        tags = np.zeros((h, w), np.uint32)
        base, _ = O.render(threads=threads, tags=tags)
        unstable = np.zeros((h, w), bool)
        for px, tol in [(p, STABLE_TOL + GRADIENT * p) for p in JITTER_PX] + [(p, JUMP_TOL) for p in JUMP_PX]:
            dj = px / h
            for jx, jy in itertools.product((-dj, 0.0, dj), repeat=2):
                if jx == 0.0 and jy == 0.0:
                    continue
                ij, _ = O.render(threads=threads, jitter=(jx, jy))
                unstable |= _diff(ij, base) > tol
        _PROBES[key] = (base, tags, unstable, ref["frame"])     # (the frame is kept so that its id stays unique)
    base, tags, unstable, _ = _PROBES[key]
    img = base if candidate is None else candidate
    d = _diff(img, ref["frame"])
    bad = d > TOL
    out = dict(pixels=w * h, over=int(bad.sum()), max=float(d.max()))
    left = bad & ~unstable
    out["unstable"] = int((bad & unstable).sum())
    out["unstable_pixels_in_frame"] = int(unstable.sum())

    def claim(name, mask):
        nonlocal left
        c = left & mask
        out[name] = int(c.sum())
        left = left & ~c
    claim("divergent", (tags & oracle.TAG_QUAD_DIVERGENT) != 0)
    claim("torus", ((tags & oracle.TAG_TORUS) != 0) & (d <= TORUS_TOL))
    claim("t21", ((tags & oracle.TAG_BOX_INSIDE) != 0) & (d <= T21_TOL))
    claim("box_nan", (tags & oracle.TAG_BOX_NAN) != 0)
    # a mip-mapped fetch differences the uv of its 2x2-quad neighbours: if one of those is itself unstable or divergent, so is this LOD
    flagged = unstable | ((tags & oracle.TAG_QUAD_DIVERGENT) != 0)
    hq, wq = (h // 2) * 2, (w // 2) * 2
    quad_any = np.zeros_like(flagged)
    q = flagged[:hq, :wq].reshape(hq // 2, 2, wq // 2, 2).any(axis=(1, 3))
    quad_any[:hq, :wq] = np.repeat(np.repeat(q, 2, axis=0), 2, axis=1)
    claim("quad_neighbour", ((tags & oracle.TAG_TEXTURE) != 0) & quad_any)
    claim("texture", ((tags & oracle.TAG_TEXTURE) != 0) & (d <= tex_tol))
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
    claim("approx_math", d <= APPROX_TOL)
    out["unexplained"] = int(left.sum())
    out["where"] = [(int(x), int(y), float(d[y, x]), int(tags[y, x])) for y, x in np.argwhere(left)[:8]]
    return out
