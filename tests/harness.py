"""ctypes binding of tests/host_harness (the product's device header compiled for the host).

Test infrastructure only -- lets `-m "not gpu"` tests compare the tracer's LOGIC with the oracle.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "host_harness")
_LIB = os.path.join(_DIR, "libharness.so")
TEX_SLOTS = ("texture_sphere_1", "texture_sphere_2", "texture_sphere_3", "texture_sphere_4", "texture_ring", "texture_box")
BLOCKS = ("scene_buf", "spheres_buf", "planes_buf", "surfaces_buf", "boxes_buf", "toruses_buf", "rings_buf", "lights_point_buf",
          "lights_direct_buf")


class Defines(ctypes.Structure):
    _fields_ = [(f"i{k}", ctypes.c_int32) for k in range(9)] + [("ambient", ctypes.c_float * 3), ("shadow", ctypes.c_float * 3)]


class Tex(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("channels", ctypes.c_int32), ("wrap", ctypes.c_int32),
                ("texels", ctypes.c_void_p)]


class Frame(ctypes.Structure):
    _fields_ = [("fb_width", ctypes.c_int32), ("fb_height", ctypes.c_int32), ("defines", Defines), ("blocks", ctypes.c_void_p * 9),
                ("block_sizes", ctypes.c_uint64 * 9), ("sky_size", ctypes.c_int32), ("sky_channels", ctypes.c_int32),
                ("sky_faces", ctypes.c_void_p * 6), ("tex", Tex * 6), ("cull", ctypes.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.run(["make", "-C", _DIR], check=True, stdout=subprocess.DEVNULL)
        l = ctypes.CDLL(_LIB)
        l.harness_render.restype = ctypes.c_int
        l.harness_render.argtypes = [ctypes.POINTER(Frame), ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        l.harness_unorm8_mismatches.restype = ctypes.c_int
        l.harness_kat.restype = ctypes.c_int
        l.harness_kat.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_float * 3, ctypes.c_float * 3, ctypes.c_float, ctypes.c_float * 3]
        _lib = l
    return _lib


def _frame(scene_blocks, fb_w, fb_h, textures, cubemap, cull):
    keep = []
    fr = Frame()
    fr.fb_width, fr.fb_height = fb_w, fb_h
    d = scene_blocks.defines
    for k in range(9):
        setattr(fr.defines, f"i{k}", int(d[k]))
    fr.defines.ambient = (ctypes.c_float * 3)(*d[9:12])
    fr.defines.shadow = (ctypes.c_float * 3)(*d[12:15])
    for b, name in enumerate(BLOCKS):
        raw = scene_blocks.blocks.get(name, b"")
        buf = ctypes.create_string_buffer(raw, max(len(raw), 1))
        keep.append(buf)
        fr.blocks[b] = ctypes.cast(buf, ctypes.c_void_p)
        fr.block_sizes[b] = len(raw)
    for uniform, _unit, img in (textures or ()):
        arr = np.ascontiguousarray(img, dtype=np.uint8)
        keep.append(arr)
        fr.tex[TEX_SLOTS.index(uniform)] = Tex(arr.shape[1], arr.shape[0], 1 if arr.ndim == 2 else arr.shape[2], 0, arr.ctypes.data)
    if cubemap is not None:
        faces = [None if f is None else np.ascontiguousarray(f, dtype=np.uint8) for f in cubemap]
        keep.append(faces)
        first = next(f for f in faces if f is not None)
        fr.sky_size, fr.sky_channels = first.shape[0], first.shape[2]
        for i, f in enumerate(faces):
            fr.sky_faces[i] = None if f is None else f.ctypes.data
    fr.cull = int(cull) if cull in (0, 1, 2) else (1 if cull else 0)   # 1 / True: all culls; 2: culls without the ray pencils; 0: none
    return fr, keep


def pencil_stats(scene_blocks, fb_w=64, fb_h=64):
    """The scene's ray pencils as the packer lays them out and the builder fills them: {'pencils', 'stride', 'each': [(kind, cells, mean
    set bits per cell)]}."""
    fr, _keep = _frame(scene_blocks, fb_w, fb_h, None, None, 1)
    out = (ctypes.c_double * 32)()
    n = lib().harness_pencil_stats(ctypes.byref(fr), out, 32)
    if n < 2:
        raise RuntimeError("harness_pencil_stats failed")
    return {"pencils": int(out[0]), "stride": int(out[1]), "each": [(int(out[k]), int(out[k + 1]), out[k + 2]) for k in range(2, n, 3)]}


def table_premise(scene_blocks, n_rays, seed=1):
    """Candidate tables primitive by primitive (harness_table_premise): {'rays', 'hits', 'violations', 'mean_bits', 'all_ones'} or None
    when the scene has no tables."""
    fr, _keep = _frame(scene_blocks, 64, 64, None, None, 1)
    cnt = (ctypes.c_int64 * 5)()
    fn = lib().harness_table_premise
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_int64)]
    rc = fn(ctypes.byref(fr), n_rays, seed, cnt)
    if rc == -2:
        return None
    if rc != 0:
        raise RuntimeError("harness_table_premise failed")
    return {"rays": cnt[0], "hits": cnt[1], "violations": cnt[2], "mean_bits": cnt[3] / max(1, cnt[0]), "all_ones": cnt[4]}


def render(scene_blocks, fb_w, fb_h, textures=None, cubemap=None, cull=True, y0=0, y1=None):
    fr, _keep = _frame(scene_blocks, fb_w, fb_h, textures, cubemap, cull)
    y1 = fb_h if y1 is None else y1
    out = np.empty((y1 - y0, fb_w, 4), dtype=np.float32)
    cnt = (ctypes.c_uint64 * 4)()
    rc = lib().harness_render(ctypes.byref(fr), y0, y1, out.ctypes.data, cnt)
    if rc != 0:
        raise RuntimeError("harness_render failed")
    return out, {"closest": cnt[0], "shadow_ref": cnt[1], "shadow_cast": cnt[2], "torus_solves": cnt[3]}


def probe(scene_blocks, rays):
    """Single rays through the product's own scans (harness_probe). rays: (n, 8) float32 -- ro, rd, limit, torus index. Returns (n, 12):
    literal torus hit, t | product's torus composition hit, t | in_shadow culls on, off | calc_inter culls on: t, num, type | culls off."""
    fr, _keep = _frame(scene_blocks, 64, 64, None, None, 1)
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
    out = np.zeros((rays.shape[0], 12), dtype=np.float32)
    fn = lib().harness_probe
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    if fn(ctypes.byref(fr), rays.ctypes.data, rays.shape[0], out.ctypes.data) != 0:
        raise RuntimeError("harness_probe failed")
    return out


def kat(type_, record: bytes, ro, rd, tmin=1e6):
    """(hit, t, culled) from the DEVICE intersector + its cull predicate for one std140 record."""
    out = (ctypes.c_float * 3)()
    buf = ctypes.create_string_buffer(record, len(record))
    rc = lib().harness_kat(type_, buf, (ctypes.c_float * 3)(*ro), (ctypes.c_float * 3)(*rd), tmin, out)
    if rc != 0:
        raise RuntimeError(f"harness_kat failed ({rc})")
    return bool(out[0]), out[1], bool(out[2])


def smaa(color_rgba8, preset, area, search, planes: bool = True, roles: bool = False):
    """The product's SMAA arithmetic (csrc/smaa_device.h, host build) run densely: {'edges', 'blend', 'screen'} like oracle.smaa.run.
    planes: the orthogonal searches count their steps on the bit planes, like the HIP weight kernel (False: per-step loops only).
    roles: a pixel's weights composed from its independent parts the way the role-split HIP kernel composes them (four waves, one part each)."""
    color = np.ascontiguousarray(color_rgba8, np.uint8)
    h, w = color.shape[:2]
    area, search = np.ascontiguousarray(area, np.uint8), np.ascontiguousarray(search, np.uint8)
    edges, blend, screen = np.empty((h, w, 2), np.uint8), np.empty((h, w, 4), np.uint8), np.empty((h, w, 4), np.uint8)
    l = lib()
    l.harness_smaa_use_planes(1 if planes else 0)
    l.harness_smaa_use_roles(1 if roles else 0)
    l.harness_smaa.restype = ctypes.c_int
    l.harness_smaa.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5
    p = ("LOW", "MEDIUM", "HIGH", "ULTRA").index(preset) if isinstance(preset, str) else int(preset)
    if l.harness_smaa(color.ctypes.data, w, h, p, area.ctypes.data, search.ctypes.data, edges.ctypes.data, blend.ctypes.data, screen.ctypes.data) != 0:
        raise RuntimeError("harness_smaa failed")
    return {"edges": edges, "blend": blend, "screen": screen}
