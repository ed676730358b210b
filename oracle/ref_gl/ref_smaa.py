"""The reference's SMAA post-process (GLWrapper.cpp:173-204) on Mesa llvmpipe, head-less -- TEST INFRASTRUCTURE, build
container only, like ref_gl.py.

Nothing of the reference is copied: the shader library (assets/shaders/SMAA.h), the six little main() wrappers and the GLSL
header (raw string literals inside src/SMAA_Builder.h) and the two look-up tables (byte arrays inside src/AreaTex.h and
src/SearchTex.h) are READ from /root/reference at run time and assembled exactly as SMAA_Builder does
(SMAA_Builder.h:17-33,87-113). The three passes then run the way GLWrapper::draw runs them: colour (RGBA8) -> edges (RG8,
cleared to 0, `discard` leaves 0) -> blend weights (RGBA8) -> screen (RGBA8), every texture LINEAR + CLAMP_TO_EDGE."""
from __future__ import annotations

import ctypes
import os
import re

import numpy as np

from . import ref_gl

PRESETS = ("LOW", "MEDIUM", "HIGH", "ULTRA")     # enum SMAA_PRESET, SMAA_Builder.h:9-12


def _builder_text() -> str:
    return open(os.path.join(ref_gl.REFERENCE_ROOT, "src", "SMAA_Builder.h")).read()


def _raw_string(text: str, name: str) -> str:
    m = re.search(r"const\s+std::string\s+" + re.escape(name) + r'\s*=\s*R"X\((.*?)\)X"', text, re.S)
    if not m:
        raise RuntimeError(f"SMAA_Builder.h: string '{name}' not found")
    return m.group(1)


def programs(width: int, height: int, preset: str) -> dict:
    """{'edge': (vs, ps), 'blend': (vs, ps), 'neighborhood': (vs, ps)} as SMAA_Builder assembles them."""
    t = _builder_text()
    body = open(os.path.join(ref_gl.REFERENCE_ROOT, "assets", "shaders", "SMAA.h")).read()
    header = _raw_string(t, "glsl_header") + "\n#define SMAA_RT_METRICS float4(1.0 / %d.0, 1.0 / %d.0, %d.0, %d.0)\n#define SMAA_PRESET_%s" % (
        width, height, width, height, preset)                                   # SMAA_Builder.h:31-33
    hv, hp = _raw_string(t, "header_vs"), _raw_string(t, "header_ps")
    return {k: (header + hv + body + _raw_string(t, k + "_vs"), header + hp + body + _raw_string(t, k + "_ps")) for k in ("edge", "blend", "neighborhood")}


def _byte_array(path: str, array_name: str) -> np.ndarray:
    text = open(path).read()
    m = re.search(re.escape(array_name) + r"\s*\[\s*\]\s*=\s*\{(.*?)\}\s*;", text, re.S)
    if not m:
        raise RuntimeError(f"{path}: array {array_name} not found")
    return np.array([int(v, 0) for v in re.findall(r"0x[0-9a-fA-F]+|\d+", m.group(1))], dtype=np.uint8)


def reference_luts():
    """(area 560x160x2 uint8, search 16x64 uint8) parsed from the reference's AreaTex.h / SearchTex.h (SMAA_Builder.h:52-83)."""
    src = os.path.join(ref_gl.REFERENCE_ROOT, "src")
    area = _byte_array(os.path.join(src, "AreaTex.h"), "areaTexBytes")
    search = _byte_array(os.path.join(src, "SearchTex.h"), "searchTexBytes")
    assert area.size == 160 * 560 * 2 and search.size == 64 * 16
    return area.reshape(560, 160, 2), search.reshape(16, 64)


class _PassTex(ctypes.Structure):
    _fields_ = [("uniform_name", ctypes.c_char_p), ("unit", ctypes.c_int), ("width", ctypes.c_int), ("height", ctypes.c_int),
                ("channels", ctypes.c_int), ("texels", ctypes.c_void_p)]


def _pass(vs: str, ps: str, w: int, h: int, inputs, out_channels: int) -> np.ndarray:
    l = ref_gl.lib()
    l.glref_pass.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_PassTex), ctypes.c_int, ctypes.c_void_p]
    keep = []
    arr = (_PassTex * len(inputs))()
    for k, (name, unit, img) in enumerate(inputs):
        a = np.ascontiguousarray(img, np.uint8)
        keep.append(a)
        ch = 1 if a.ndim == 2 else a.shape[2]
        arr[k] = _PassTex(name.encode(), unit, a.shape[1], a.shape[0], ch, a.ctypes.data)
    out = np.empty((h, w, out_channels), np.uint8)
    if l.glref_pass(vs.encode(), ps.encode(), w, h, len(inputs), arr, out_channels, out.ctypes.data) != 0:
        raise RuntimeError("glref_pass: " + l.glref_error().decode())
    return out


def run(color_rgba8: np.ndarray, preset: str = "ULTRA", area=None, search=None) -> dict:
    """color_rgba8: (H, W, 4) uint8, row 0 = bottom row (texture row 0). Returns {'edges': (H,W,2), 'blend': (H,W,4),
    'screen': (H,W,4)} uint8 -- what fboTexEdge, fboTexBlend and the default framebuffer hold after GLWrapper::draw."""
    h, w = color_rgba8.shape[:2]
    if area is None or search is None:
        area, search = reference_luts()
    p = programs(w, h, preset)
    edges = _pass(*p["edge"], w, h, [("color_tex", 0, color_rgba8)], 2)                                                         # GLWrapper.cpp:173-180
    blend = _pass(*p["blend"], w, h, [("edge_tex", 0, edges), ("area_tex", 1, area), ("search_tex", 2, search)], 4)            # :182-193
    screen = _pass(*p["neighborhood"], w, h, [("color_tex", 0, color_rgba8), ("blend_tex", 1, blend)], 4)                       # :195-204
    return {"edges": edges, "blend": blend, "screen": screen}
