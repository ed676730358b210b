"""The two SMAA look-up tables for tests, smoke and bench (the counterpart of textures.py for SMAA).

The reference uploads two third-party byte tables (src/AreaTex.h 160x560 RG8, src/SearchTex.h 64x16 R8; SMAA_Builder.h:52-83). This
repository stores neither; both are computed from their published construction by include/rtx/smaa_tables.h inside librtx_hip.so
(rtx_smaa_default_tables -- host code, needs no GPU), and tests/test_smaa_tables.py pins the result: byte-identical to the reference's
arrays where /root/reference exists, sha256 everywhere.

  area_table()            -- (560, 160, 2) uint8, == areaTexBytes. What rtx_enable_smaa uses when the caller supplies nothing.
  search_table()          -- (16, 64) uint8, == searchTexBytes; computed HERE in Python from the same definition, independently of the
                             C++ generator (the two are compared in the test).
  synthetic_area_table()  -- an area table of the right shape with made-up contents (unsmoothed trapezoids, a smooth invented function for
                             the diagonals): the three passes are right or wrong independently of what the tables hold, and parity tests
                             also run with this one so that they do not depend on the real table's many zero entries.
"""
from __future__ import annotations

import itertools

import numpy as np


def _bilinear(e):
    """value of a bilinear fetch at (-0.25, -0.125) over edges e = (e0, e1, e2, e3) in {0,1}"""
    lerp = lambda a, b, p: a + (b - a) * p
    a = lerp(e[0], e[1], 1.0 - 0.25)
    b = lerp(e[2], e[3], 1.0 - 0.25)
    return lerp(a, b, 1.0 - 0.125)


def search_table() -> np.ndarray:
    """(16, 64) uint8."""
    edge = {_bilinear(e): e for e in itertools.product((0, 1), repeat=4)}

    def delta_left(left, top):
        d = 0
        if top[3] == 1:                      # there is an edge: continue
            d += 1
        if d == 1 and top[2] == 1 and left[1] != 1 and left[3] != 1:   # another edge and no crossing edges: continue
            d += 1
        return d

    def delta_right(left, top):
        d = 0
        if top[3] == 1 and left[1] != 1 and left[3] != 1:
            d += 1
        if d == 1 and top[2] == 1 and left[0] != 1 and left[2] != 1:
            d += 1
        return d

    img = np.zeros((33, 66), np.uint8)       # [y, x]
    for x in range(33):
        for y in range(33):
            tx, ty = 0.03125 * x, 0.03125 * y
            if tx in edge and ty in edge:
                img[y, x] = 127 * delta_left(edge[tx], edge[ty])
                img[y, 33 + x] = 127 * delta_right(edge[tx], edge[ty])
    img = img[17:33, 0:64]                   # crop to 64 x 16 ...
    return np.ascontiguousarray(img[::-1])   # ... and flip vertically


def area_table() -> np.ndarray:
    """(560, 160, 2) uint8: the library's generated table (== the reference's areaTexBytes)."""
    from . import _capi
    out = np.zeros((560, 160, 2), np.uint8)
    st = _capi.load().rtx_smaa_default_tables(out.ctypes.data, out.nbytes, None, 0)
    if st != _capi.RTX_OK:
        raise RuntimeError("rtx_smaa_default_tables failed")
    return out


def library_search_table() -> np.ndarray:
    """(16, 64) uint8: the C++ generator's search table (compared with search_table() in tests/test_smaa_tables.py)."""
    from . import _capi
    out = np.zeros((16, 64), np.uint8)
    st = _capi.load().rtx_smaa_default_tables(None, 0, out.ctypes.data, out.nbytes)
    if st != _capi.RTX_OK:
        raise RuntimeError("rtx_smaa_default_tables failed")
    return out


def synthetic_area_table(seed: int = 7) -> np.ndarray:
    """(560, 160, 2) uint8, synthetic (see the module docstring)."""
    t = np.zeros((560, 160, 2), np.float64)
    # orthogonal half: 5 x 5 blocks of 16 x 16 (block index = round(4 e), values 0,1,3,4; 2 never occurs); texel i <-> distance i^2
    height = {0: 0.0, 1: -0.5, 3: 0.5, 4: 0.0, 2: 0.0}
    for p1, p2 in itertools.product(range(5), repeat=2):
        h1, h2 = height[p1], -height[p2]
        for i, j in itertools.product(range(16), repeat=2):
            d1, d2 = float(i * i), float(j * j)
            L = d1 + d2 + 1.0
            if h1 != 0.0 and h2 != 0.0 and h1 * h2 > 0:        # U shape: two ramps meeting at 0 in the middle
                f = lambda x: h1 * max(0.0, 1.0 - 2.0 * x / L) + h2 * max(0.0, 2.0 * x / L - 1.0)
            else:                                              # Z / L shape: one straight line
                f = lambda x: h1 + (h2 - h1) * x / L
            xs = d1 + (np.arange(8) + 0.5) / 8.0
            v = np.array([f(x) for x in xs])
            a_up, a_dn = np.clip(v, 0, None).mean(), np.clip(-v, 0, None).mean()
            for sub in range(7):
                shift = (0.0, -0.25, 0.25, -0.125, 0.125, -0.375, 0.375)[sub]
                t[sub * 80 + p2 * 16 + j, p1 * 16 + i] = (np.clip(a_up + shift * (a_up > 0), 0, 1), np.clip(a_dn - shift * (a_dn > 0), 0, 1))
    # diagonal half: 4 x 4 blocks of 20 x 20 at column 80
    rng = np.random.default_rng(seed)
    k = rng.uniform(0.6, 1.0, (4, 4, 2))
    for e1, e2 in itertools.product(range(4), repeat=2):
        if e1 == 0 and e2 == 0:
            continue
        for i, j in itertools.product(range(20), repeat=2):
            L = i + j + 1.0
            pos = (i + 0.5) / L
            a = 0.5 * (1.0 - abs(2.0 * pos - 1.0)) if (e1 and e2) else 0.5 * (1.0 - pos if e1 else pos)
            for sub in range(7):
                t[sub * 80 + e2 * 20 + j, 80 + e1 * 20 + i] = (a * k[e1, e2, 0], a * k[e1, e2, 1] * (0.3 + 0.1 * sub))
    return np.ascontiguousarray(np.clip(t * 255.0 + 0.5, 0, 255).astype(np.uint8))
