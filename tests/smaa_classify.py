"""Pixel-by-pixel accounting of the differences between the SMAA oracle and a real GL implementation (Mesa llvmpipe running
the reference's own SMAA.h, tests/golden/smaa_ref_*.npz).

GL leaves two things to the implementation that SMAA's branches are sensitive to: how precisely the texture-coordinate varyings
are interpolated and how precisely bilinear weights are formed. The oracle (and the HIP kernels) evaluate the shaders with EXACT
pixel positions (oracle/smaa_oracle.c); llvmpipe carries ~1e-5 texels of float noise. Three places turn that noise into
different bytes, and each is recognisable per pixel:

 pass 1  a luma delta within float rounding of the threshold / of the local-contrast test (SMAA.h:714,737): the edge bit may flip;
 pass 2  (a) `e.g > 0.0` / `e.r > 0.0` on a LINEAR fetch at the pixel centre (SMAA.h:1152,1155,1205): with noise the fetch picks up
             1e-5 of a NEIGHBOUR's edge, and a pixel without an edge runs the whole weight computation ("phantom edge");
         (b) after SMAA_MAX_SEARCH_STEPS steps `texcoord > end` (SMAA.h:1029,1057...) compares two numbers that are equal in
             exact arithmetic: one more search step or not;
     both are reproduced by oracle.smaa.blend_pass_jitter (positions displaced by +-1e-5 / +-1e-3 texels, each search end by +-1e-3);
     a phantom pixel that no uniform displacement reproduces must EQUAL what oracle.smaa.blend_pass_forced computes for it -- exact
     positions, the branch taken although the pixel has no edge of its own -- it is not excused unseen;
 pass 3  `max(a.x, a.z) > max(a.y, a.w)` (SMAA.h:1279) on EQUAL bytes -- common for diagonals -- is decided by the noise.
Everything else must agree to 1 LSB (bilinear weight rounding)."""
from __future__ import annotations

import itertools

import numpy as np

from oracle import smaa

JITTERS = (-1e-3, -1e-5, 0.0, 1e-5, 1e-3)
SLACKS = (0.0, 1e-3, -1e-3)


def classify_blend(ref_edges, ref_blend, preset, area, search):
    """Pass 2 on the REFERENCE's edge texture. Returns dict(differing, explained_by_noise, phantom, unexplained) pixel counts;
    `unexplained` must be 0."""
    ours = smaa.blend_pass(ref_edges, preset, area, search).astype(np.int16)
    ref = ref_blend.astype(np.int16)
    pairs = (slice(0, 2), slice(2, 4))
    bad = [(np.abs(ref[..., p] - ours[..., p]) > 1).any(-1) for p in pairs]
    expl = [np.zeros_like(bad[0]), np.zeros_like(bad[0])]
    if bad[0].any() or bad[1].any():
        for jx, jy in itertools.product(JITTERS, repeat=2):
            for lo, hi in itertools.product(SLACKS, repeat=2):
                bj = smaa.blend_pass_jitter(ref_edges, preset, area, search, jx, jy, lo, hi).astype(np.int16)
                for k, p in enumerate(pairs):
                    expl[k] |= (np.abs(ref[..., p] - bj[..., p]) <= 1).all(-1)
    own_g0, own_r0 = ref_edges[..., 1] == 0, ref_edges[..., 0] == 0
    # phantom: the reference computed north weights (rg) for a pixel whose own north edge is 0 -- then its west pair is whatever the
    # "diagonal found, skip vertical" rule left -- or west weights (ba) for a pixel whose own west edge is 0
    phantom_n = own_g0 & (ref[..., 0:2] != 0).any(-1)
    phantom_w = own_r0 & (ref[..., 2:4] != 0).any(-1)
    left = [(bad[0] & ~expl[0]), (bad[1] & ~expl[1])]
    # a phantom pixel is not excused, its VALUE is demanded: what the weight computation yields at exact positions when the branch is
    # taken although the pixel has no edge of its own (the searches, area look-ups and corner rounding see the real edge texture)
    forced = [np.zeros_like(bad[0]), np.zeros_like(bad[0])]
    if ((left[0] | left[1]) & (phantom_n | phantom_w)).any():
        for force in (1, 2, 3):
            for lo, hi in itertools.product(SLACKS, repeat=2):
                bf = smaa.blend_pass_forced(ref_edges, preset, area, search, force, lo, hi).astype(np.int16)
                for k, p in enumerate(pairs):
                    forced[k] |= (np.abs(ref[..., p] - bf[..., p]) <= 1).all(-1)
    left = [left[0] & ~forced[0], left[1] & ~forced[1]]
    phantom = ((bad[0] & ~expl[0] & forced[0]) | (bad[1] & ~expl[1] & forced[1])) & (phantom_n | phantom_w)
    unexplained = left[0] | left[1]
    return dict(differing=int((bad[0] | bad[1]).sum()), explained_by_noise=int(((bad[0] & expl[0]) | (bad[1] & expl[1])).sum()),
                phantom=int(phantom.sum()), unexplained=int(unexplained.sum()), where=np.argwhere(unexplained)[:8].tolist())


def classify_neighborhood(color, ref_blend, ref_screen):
    """Pass 3 on the REFERENCE's weight texture: differences beyond 1 LSB only at exact ties of the horizontal / vertical maxima."""
    ours = smaa.neighborhood_pass(color, ref_blend).astype(np.int16)
    d = np.abs(ours - ref_screen.astype(np.int16)).max(-1)
    B = ref_blend.astype(np.int16)
    Bp = np.pad(B, ((0, 1), (0, 1), (0, 0)), mode="edge")
    ax, ay, aw, az = Bp[:-1, 1:, 3], Bp[1:, :-1, 1], B[..., 0], B[..., 2]
    tie = (np.maximum(ax, az) == np.maximum(ay, aw)) & ((ax + ay + az + aw) > 0)
    return dict(differing=int((d > 1).sum()), at_ties=int(((d > 1) & tie).sum()), unexplained=int(((d > 1) & ~tie).sum()),
                one_lsb=int((d == 1).sum()), where=np.argwhere((d > 1) & ~tie)[:8].tolist())


def classify_edges(color, ref_edges, preset):
    """Pass 1: an edge bit may differ only where a luma delta is within 1e-5 (relative) of a decision."""
    ours = smaa.run(color, preset, np.zeros(smaa.AREA_SHAPE, np.uint8), np.zeros(smaa.SEARCH_SHAPE, np.uint8))["edges"]
    diff = (ours != ref_edges).any(-1)
    n_bad = 0
    if diff.any():
        thr = (0.15, 0.1, 0.1, 0.05)[smaa.PRESETS.index(preset) if isinstance(preset, str) else preset]
        L = (color[..., 0].astype(np.float64) * 0.2126 + color[..., 1].astype(np.float64) * 0.7152 + color[..., 2].astype(np.float64) * 0.0722) / 255.0
        Lp = np.pad(L, 2, mode="edge")
        for y, x in np.argwhere(diff):
            c = Lp[y + 2, x + 2]
            nb = dict(l=Lp[y + 2, x + 1], t=Lp[y + 1, x + 2], r=Lp[y + 2, x + 3], b=Lp[y + 3, x + 2], ll=Lp[y + 2, x], tt=Lp[y, x + 2])
            deltas = [abs(c - nb["l"]), abs(c - nb["t"]), abs(c - nb["r"]), abs(c - nb["b"]), abs(nb["l"] - nb["ll"]), abs(nb["t"] - nb["tt"])]
            near_thr = any(abs(dl - thr) < 1e-6 for dl in deltas[:2])
            fin = max(deltas)
            near_lca = any(abs(fin - 2.0 * dl) < 1e-6 for dl in deltas[:2])
            if not (near_thr or near_lca):
                n_bad += 1
    return dict(differing=int(diff.sum()), unexplained=n_bad)
