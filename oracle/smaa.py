"""ctypes binding of oracle/smaa_oracle.c -- the CPU restatement of the reference's SMAA post-process. TEST INFRASTRUCTURE
(the checker): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it."""
from __future__ import annotations

import ctypes

import numpy as np

from . import oracle

PRESETS = ("LOW", "MEDIUM", "HIGH", "ULTRA")   # reference: enum SMAA_PRESET, src/SMAA_Builder.h:9-12
AREA_SHAPE, SEARCH_SHAPE = (560, 160, 2), (16, 64)


def _lib():
    l = oracle.lib()
    l.smaa_oracle_run.restype = ctypes.c_int
    l.smaa_oracle_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_void_p, ctypes.c_void_p]
    l.smaa_oracle_blend_pass.restype = ctypes.c_int
    l.smaa_oracle_blend_pass.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    l.smaa_oracle_neighborhood_pass.restype = ctypes.c_int
    l.smaa_oracle_neighborhood_pass.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return l


def _preset(p) -> int:
    return PRESETS.index(p) if isinstance(p, str) else int(p)


def _luts(area, search):
    area = np.ascontiguousarray(area, np.uint8)
    search = np.ascontiguousarray(search, np.uint8)
    assert area.shape == AREA_SHAPE and search.shape == SEARCH_SHAPE, (area.shape, search.shape)
    return area, search


def run(color_rgba8: np.ndarray, preset, area, search) -> dict:
    """color_rgba8 (H, W, 4) uint8, row 0 = bottom row. Returns {'edges': (H,W,2), 'blend': (H,W,4), 'screen': (H,W,4)}."""
    color = np.ascontiguousarray(color_rgba8, np.uint8)
    h, w = color.shape[:2]
    area, search = _luts(area, search)
    edges, blend, screen = np.empty((h, w, 2), np.uint8), np.empty((h, w, 4), np.uint8), np.empty((h, w, 4), np.uint8)
    rc = _lib().smaa_oracle_run(color.ctypes.data, w, h, _preset(preset), area.ctypes.data, search.ctypes.data, edges.ctypes.data, blend.ctypes.data,
                                screen.ctypes.data)
    if rc != 0:
        raise RuntimeError("smaa_oracle_run failed")
    return {"edges": edges, "blend": blend, "screen": screen}


def blend_pass(edges_rg8: np.ndarray, preset, area, search) -> np.ndarray:
    edges = np.ascontiguousarray(edges_rg8, np.uint8)
    h, w = edges.shape[:2]
    area, search = _luts(area, search)
    out = np.empty((h, w, 4), np.uint8)
    if _lib().smaa_oracle_blend_pass(edges.ctypes.data, w, h, _preset(preset), area.ctypes.data, search.ctypes.data, out.ctypes.data) != 0:
        raise RuntimeError("smaa_oracle_blend_pass failed")
    return out


def blend_pass_jitter(edges_rg8: np.ndarray, preset, area, search, jx: float, jy: float, slack_lo: float = 0.0, slack_hi: float = 0.0) -> np.ndarray:
    """Diagnostic: pass 2 with every pixel's position displaced by (jx, jy) texels (see smaa_oracle.c)."""
    edges = np.ascontiguousarray(edges_rg8, np.uint8)
    h, w = edges.shape[:2]
    area, search = _luts(area, search)
    out = np.empty((h, w, 4), np.uint8)
    l = _lib()
    l.smaa_oracle_blend_pass_jitter.restype = ctypes.c_int
    l.smaa_oracle_blend_pass_jitter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
    if l.smaa_oracle_blend_pass_jitter(edges.ctypes.data, w, h, _preset(preset), area.ctypes.data, search.ctypes.data, jx, jy, slack_lo, slack_hi, out.ctypes.data) != 0:
        raise RuntimeError("smaa_oracle_blend_pass_jitter failed")
    return out


def blend_pass_forced(edges_rg8: np.ndarray, preset, area, search, force: int, slack_lo: float = 0.0, slack_hi: float = 0.0) -> np.ndarray:
    """Diagnostic: pass 2 at exact positions with "phantom" north (force & 1) / west (force & 2) edges taken (see smaa_oracle.c)."""
    edges = np.ascontiguousarray(edges_rg8, np.uint8)
    h, w = edges.shape[:2]
    area, search = _luts(area, search)
    out = np.empty((h, w, 4), np.uint8)
    l = _lib()
    l.smaa_oracle_blend_pass_forced.restype = ctypes.c_int
    l.smaa_oracle_blend_pass_forced.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
    if l.smaa_oracle_blend_pass_forced(edges.ctypes.data, w, h, _preset(preset), area.ctypes.data, search.ctypes.data, int(force), slack_lo, slack_hi, out.ctypes.data) != 0:
        raise RuntimeError("smaa_oracle_blend_pass_forced failed")
    return out


def neighborhood_pass(color_rgba8: np.ndarray, blend_rgba8: np.ndarray) -> np.ndarray:
    color, blend = np.ascontiguousarray(color_rgba8, np.uint8), np.ascontiguousarray(blend_rgba8, np.uint8)
    h, w = color.shape[:2]
    out = np.empty((h, w, 4), np.uint8)
    if _lib().smaa_oracle_neighborhood_pass(color.ctypes.data, blend.ctypes.data, w, h, out.ctypes.data) != 0:
        raise RuntimeError("smaa_oracle_neighborhood_pass failed")
    return out
