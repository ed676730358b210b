"""ctypes driver of oracle/ref_gl/glref.c: the reference's OWN fragment shader on Mesa llvmpipe, head-less.

TEST INFRASTRUCTURE, build container only: the shader text is read from /root/reference at run time (it is never
copied into this repository) and templated the way GLWrapper::init_shaders does (GLWrapper.cpp:232-277).
Used by tools/gen_reference_frames.py to produce tests/golden/ref_frame_*.npz, the vectors that pin the oracle
against the reference itself."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(os.path.dirname(_HERE), "_ref", "libglref.so")
REFERENCE_ROOT = os.environ.get("RT_REFERENCE_ROOT", "/root/reference")
BLOCKS = ("scene_buf", "spheres_buf", "planes_buf", "surfaces_buf", "boxes_buf", "toruses_buf", "rings_buf",
          "lights_point_buf", "lights_direct_buf")   # binding point = index (SceneManager.cpp:244-256)


class _Tex(ctypes.Structure):
    _fields_ = [("uniform_name", ctypes.c_char_p), ("unit", ctypes.c_int), ("width", ctypes.c_int), ("height", ctypes.c_int),
                ("channels", ctypes.c_int), ("clamp_to_edge", ctypes.c_int), ("texels", ctypes.c_void_p),
                ("n_levels", ctypes.c_int), ("levels", ctypes.POINTER(ctypes.c_void_p))]


class _Cube(ctypes.Structure):
    _fields_ = [("size", ctypes.c_int), ("channels", ctypes.c_int), ("gen_mipmap", ctypes.c_int), ("faces", ctypes.c_void_p * 6)]


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "assets", "shaders")) and os.path.exists("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so")


_lib = None


def lib():
    global _lib
    if _lib is None:
        os.makedirs(os.path.dirname(_LIB), exist_ok=True)
        src = os.path.join(_HERE, "glref.c")
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
            subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-o", _LIB, src, "-ldl"], check=True)
        l = ctypes.CDLL(_LIB)
        l.glref_error.restype = ctypes.c_char_p
        l.glref_renderer.restype = ctypes.c_char_p
        l.glref_version.restype = ctypes.c_char_p
        if l.glref_init(b"") != 0:
            raise RuntimeError("glref_init: " + l.glref_error().decode())
        _lib = l
    return _lib


def renderer() -> str:
    l = lib()
    return f"{l.glref_renderer().decode()} / GL {l.glref_version().decode()}"


def _f(v: float) -> str:
    return "%f" % v   # std::to_string(float)


def shader_sources(defines) -> tuple[str, str]:
    """quad.vert and rt.frag of the reference, the latter with its {NAME} placeholders replaced (first occurrence
    only, like the reference's replace())."""
    sh = os.path.join(REFERENCE_ROOT, "assets", "shaders")
    vert = open(os.path.join(sh, "quad.vert")).read()
    frag = open(os.path.join(sh, "rt.frag")).read()
    d = list(defines)
    subst = [("{SPHERE_SIZE}", str(int(d[0]))), ("{PLANE_SIZE}", str(int(d[1]))), ("{SURFACE_SIZE}", str(int(d[2]))),
             ("{BOX_SIZE}", str(int(d[3]))), ("{TORUS_SIZE}", str(int(d[4]))), ("{RING_SIZE}", str(int(d[5]))),
             ("{LIGHT_POINT_SIZE}", str(int(d[6]))), ("{LIGHT_DIRECT_SIZE}", str(int(d[7]))), ("{ITERATIONS}", str(int(d[8]))),
             ("{AMBIENT_COLOR}", "vec3(%s,%s,%s)" % (_f(d[9]), _f(d[10]), _f(d[11]))),
             ("{SHADOW_AMBIENT}", "vec3(%s,%s,%s)" % (_f(d[12]), _f(d[13]), _f(d[14])))]
    for key, val in subst:
        frag = frag.replace(key, val, 1)
    return vert, frag


def _oracle_mip_chain(arr):
    """RGBA8 mip levels as the oracle builds them (diagnostic: see glref_tex2d.levels)."""
    from oracle import oracle
    ol = oracle.lib()
    ol.orc_kat_mip_level.restype = ctypes.c_int
    ol.orc_kat_mip_level.argtypes = [ctypes.POINTER(oracle.Texture), ctypes.c_int, ctypes.c_void_p]
    ch = 1 if arr.ndim == 2 else arr.shape[2]
    ot = oracle.Texture(arr.shape[1], arr.shape[0], ch, 0, arr.ctypes.data)
    levels, L = [], 0
    while True:
        buf = np.empty(arr.shape[0] * arr.shape[1] * 4, np.uint8)
        r = ol.orc_kat_mip_level(ctypes.byref(ot), L, buf.ctypes.data)
        if r == 0:
            return levels
        levels.append(np.ascontiguousarray(buf[: (r >> 16) * (r & 0xffff) * 4]))
        L += 1


def generated_mips(img) -> list:
    """The mip levels >= 1 that llvmpipe's glGenerateMipmap builds for this image (each (h, w, 4) uint8), level 1 first."""
    l = lib()
    arr = np.ascontiguousarray(img, dtype=np.uint8)
    ch = 1 if arr.ndim == 2 else arr.shape[2]
    l.glref_generated_mip.restype = ctypes.c_int
    l.glref_generated_mip.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    out, L = [], 1
    while True:
        buf = np.empty(arr.shape[0] * arr.shape[1] * 4, np.uint8)
        r = l.glref_generated_mip(arr.shape[1], arr.shape[0], ch, arr.ctypes.data, L, buf.ctypes.data)
        if r < 0:
            raise RuntimeError("glref_generated_mip: " + l.glref_error().decode())
        if r == 0:
            return out
        w, h = r >> 16, r & 0xffff
        out.append(np.ascontiguousarray(buf[: w * h * 4].reshape(h, w, 4)))
        L += 1


def instrument_primary_hit(frag: str) -> str:
    """Run-time instrumentation of the reference's shader text (in memory only, never stored): FragColor becomes what the FIRST calcInter
    of the pixel returned -- (t, type, num, 1) -- instead of the colour. Two insertions into main() (rt.frag:804-902): globals in front of
    it, a capture right after `tm = calcInter(ro, rd, num, type);`, and the output line. Everything else -- the intersectors, the scan,
    Durand-Kerner -- is the reference's own code, which is the point: the accepted ROOT is compared, not a colour derived from it."""
    a = "void main()"
    b = "tm = calcInter(ro, rd, num, type);"
    c = "FragColor = vec4(color,1);"
    assert frag.count(a) == 1 and frag.count(b) == 1 and frag.count(c) >= 1
    frag = frag.replace(a, "float dbg_t = 0.0; int dbg_type = -1; int dbg_num = -1; bool dbg_first = true;\n" + a, 1)
    frag = frag.replace(b, b + "\n\t\tif (dbg_first) { dbg_first = false; dbg_t = tm; if (tm < maxDist) { dbg_type = type; dbg_num = num; } }", 1)
    return frag.replace(c, "FragColor = vec4(dbg_t, float(dbg_type), float(dbg_num), 1);", 1)


def render(scene_blocks, fb_w: int, fb_h: int, textures=None, cubemap=None, cube_mipmap: bool = False, oracle_mips: bool = False, level0_only: bool = False,
           patch=None):
    """One frame of the reference's program. Returns (H, W, 4) float32, row 0 = bottom row, and the set of block
    names the linked program does not contain (the reference would exit on those).
    Diagnostics: oracle_mips uploads the oracle's mip levels instead of calling glGenerateMipmap (same texels on both sides, only the
    level selection differs); level0_only uploads level 0 alone with GL_TEXTURE_MAX_LEVEL = 0, so that every fetch of the unchanged
    shader -- texture() and textureLod() alike -- is a level-0 bilinear fetch (no implementation-defined mip machinery at all)."""
    l = lib()
    vert, frag = shader_sources(scene_blocks.defines)
    if patch is not None:      # diagnostic instrumentation of the shader text, e.g. instrument_primary_hit
        frag = patch(frag)
    keep = []
    names = (ctypes.c_char_p * 9)(*[n.encode() for n in BLOCKS])
    data = (ctypes.c_void_p * 9)()
    sizes = (ctypes.c_size_t * 9)()
    for k, name in enumerate(BLOCKS):
        raw = scene_blocks.blocks.get(name, b"")
        buf = ctypes.create_string_buffer(raw, max(len(raw), 1))
        keep.append(buf)
        data[k] = ctypes.cast(buf, ctypes.c_void_p)
        sizes[k] = len(raw)
    texs = list(textures or ())
    tarr = (_Tex * max(len(texs), 1))()
    for k, (uniform, unit, img) in enumerate(texs):
        arr = np.ascontiguousarray(img, dtype=np.uint8)
        ch = 1 if arr.ndim == 2 else arr.shape[2]
        if (arr.shape[1] * ch) % 4:
            raise ValueError("row size must be a multiple of 4 bytes (the reference keeps GL_UNPACK_ALIGNMENT = 4)")
        keep.append(arr)
        nm = uniform.encode()
        keep.append(nm)
        n_levels, lv = 0, None
        if oracle_mips or level0_only:
            chain = _oracle_mip_chain(arr)
            if level0_only:
                chain = chain[:1]
            keep.append(chain)
            lv = (ctypes.c_void_p * len(chain))(*[c.ctypes.data for c in chain])
            keep.append(lv)
            n_levels = len(chain)
        tarr[k] = _Tex(nm, int(unit), arr.shape[1], arr.shape[0], ch, 0, arr.ctypes.data, n_levels, ctypes.cast(lv, ctypes.POINTER(ctypes.c_void_p)) if lv else None)
    cube = None
    if cubemap is not None:
        faces = [None if f is None else np.ascontiguousarray(f, dtype=np.uint8) for f in cubemap]
        keep.append(faces)
        first = next(f for f in faces if f is not None)
        cube = _Cube(first.shape[0], first.shape[2], 1 if cube_mipmap else 0, (ctypes.c_void_p * 6)(*[None if f is None else f.ctypes.data for f in faces]))
    out = np.empty((fb_h, fb_w, 4), dtype=np.float32)
    inactive = ctypes.c_uint(0)
    l.glref_render.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                               ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_int, ctypes.POINTER(_Tex),
                               ctypes.POINTER(_Cube), ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint)]
    rc = l.glref_render(vert.encode(), frag.encode(), fb_w, fb_h, 9, names, data, sizes, len(texs), tarr,
                        ctypes.byref(cube) if cube is not None else None, out.ctypes.data, ctypes.byref(inactive))
    if rc != 0:
        raise RuntimeError("glref_render: " + l.glref_error().decode())
    return out, {BLOCKS[k] for k in range(9) if inactive.value >> k & 1}
