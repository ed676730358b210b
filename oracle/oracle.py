"""ctypes binding of the CPU oracle (oracle/rt_oracle.c).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product
package (raytracing_opengl_amd/).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

TEX_SLOTS = ("texture_sphere_1", "texture_sphere_2", "texture_sphere_3", "texture_sphere_4", "texture_ring", "texture_box")
BLOCK_FIELDS = ("scene_buf", "spheres_buf", "planes_buf", "surfaces_buf", "boxes_buf", "toruses_buf", "rings_buf",
                "lights_point_buf", "lights_direct_buf")
TYPE_SPHERE, TYPE_PLANE, TYPE_SURFACE, TYPE_BOX, TYPE_TORUS, TYPE_RING, TYPE_POINT_LIGHT = range(7)


class Defines(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("sphere_size", "plane_size", "surface_size", "box_size", "torus_size", "ring_size",
                                              "light_point_size", "light_direct_size", "iterations")] + \
               [("ambient_color", ctypes.c_float * 3), ("shadow_ambient", ctypes.c_float * 3)]


class Texture(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("channels", ctypes.c_int32), ("wrap", ctypes.c_int32),
                ("texels", ctypes.c_void_p)]


class Cubemap(ctypes.Structure):
    _fields_ = [("face_size", ctypes.c_int32), ("channels", ctypes.c_int32), ("faces", ctypes.c_void_p * 6), ("gen_mipmap", ctypes.c_int32)]


class Frame(ctypes.Structure):
    _fields_ = [("fb_width", ctypes.c_int32), ("fb_height", ctypes.c_int32), ("defines", Defines)] + \
               [(n, ctypes.c_void_p) for n in BLOCK_FIELDS] + \
               [("skybox", Cubemap), ("tex", Texture * 6), ("texture_lod", ctypes.c_int32)]


class Counters(ctypes.Structure):
    _fields_ = [("rays_closest", ctypes.c_uint64), ("rays_shadow", ctypes.c_uint64), ("tests", ctypes.c_uint64 * 7),
                ("dk_solves", ctypes.c_uint64), ("dk_sweeps", ctypes.c_uint64), ("dk_capped", ctypes.c_uint64),
                ("t4_taken", ctypes.c_uint64), ("refract_segments", ctypes.c_uint64), ("tir_breaks", ctypes.c_uint64),
                ("alpha_pass", ctypes.c_uint64), ("side_miss", ctypes.c_uint64), ("light_hits", ctypes.c_uint64),
                ("box_nan_hits", ctypes.c_uint64), ("box_inside_hits", ctypes.c_uint64), ("segment_cap_hits", ctypes.c_uint64),
                ("max_segments", ctypes.c_uint64)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n != "tests"}
        d["tests"] = list(self.tests)
        d["rays"] = self.rays_closest + self.rays_shadow
        return d


def build(force: bool = False) -> str:
    """Compile liboracle.so with oracle/Makefile (gcc, -ffp-contract=off)."""
    newest = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("rt_oracle.c", "smaa_oracle.c"))
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < newest:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, stdout=subprocess.DEVNULL)
    return _LIB_PATH


def effective_cpus():
    """(n, quota): the CPUs this process may actually use -- the affinity mask capped by the cgroup's CPU quota -- and that quota (None if
    there is none). The GPU boxes show 256 hardware threads and grant a container 16 CPUs of time (/sys/fs/cgroup/cpu.max = "1600000 100000");
    256 OpenMP threads inside that quota are SLOWER than 16 (tools/time_oracle_threads.py: 8.3 Mray/s against 13.8 on 32)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.999)))
    return n, quota


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        l = ctypes.CDLL(_LIB_PATH)
        l.orc_render.restype = ctypes.c_int
        l.orc_render.argtypes = [ctypes.POINTER(Frame), ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(Counters), ctypes.c_int]
        l.orc_kat_intersect.restype = ctypes.c_int
        l.orc_kat_intersect.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_float * 3, ctypes.c_float * 3, ctypes.c_float, ctypes.c_int,
                                        ctypes.c_float * 5]
        l.orc_kat_atan2.restype = ctypes.c_float
        l.orc_kat_atan2.argtypes = [ctypes.c_float, ctypes.c_float]
        l.orc_kat_asin.restype = ctypes.c_float
        l.orc_kat_asin.argtypes = [ctypes.c_float]
        l.orc_kat_rotate.argtypes = [ctypes.c_float * 4, ctypes.c_float * 3, ctypes.c_float * 3]
        l.orc_kat_sample2d.argtypes = [ctypes.POINTER(Texture), ctypes.c_float, ctypes.c_float, ctypes.c_float * 4]
        l.orc_kat_sample_cube.argtypes = [ctypes.POINTER(Cubemap), ctypes.c_float * 3, ctypes.c_float * 4]
        l.orc_kat_text_round_trip.restype = ctypes.c_float
        l.orc_kat_text_round_trip.argtypes = [ctypes.c_float]
        _lib = l
    return _lib


TAG_BOX_INSIDE, TAG_REFRACT, TAG_TORUS, TAG_TEXTURE, TAG_BOX_NAN, TAG_TIR, TAG_QUAD_DIVERGENT, TAG_SKY_LOD = 1, 2, 4, 8, 16, 32, 64, 128   # ORC_TAG_* of rt_oracle.c


class OracleScene:
    """Holds one frame description (blocks + textures) alive for orc_render calls."""

    def __init__(self, scene_blocks, fb_width: int, fb_height: int, textures=None, cubemap=None, texture_lod: int = 1, cube_mipmap: bool = False):
        """scene_blocks: object with .defines (15-tuple) and .blocks (name -> bytes);
        textures: iterable of (sampler_uniform_name, unit, HxWxC uint8 array); cubemap: six NxNxC uint8 arrays;
        cube_mipmap: GLWrapper::load_cubemap(faces, genMipmap = true) (GLWrapper.cpp:307-310)."""
        self._keep = []
        fr = Frame()
        fr.fb_width, fr.fb_height = fb_width, fb_height
        d = scene_blocks.defines
        for i, (n, _t) in enumerate(Defines._fields_[:9]):
            setattr(fr.defines, n, int(d[i]))
        fr.defines.ambient_color = (ctypes.c_float * 3)(*d[9:12])
        fr.defines.shadow_ambient = (ctypes.c_float * 3)(*d[12:15])
        for name in BLOCK_FIELDS:
            raw = scene_blocks.blocks.get(name, b"")
            buf = ctypes.create_string_buffer(raw, max(len(raw), 1))
            self._keep.append(buf)
            setattr(fr, name, ctypes.cast(buf, ctypes.c_void_p))
        for uniform, _unit, img in (textures or ()):
            slot = TEX_SLOTS.index(uniform)
            arr = np.ascontiguousarray(img, dtype=np.uint8)
            self._keep.append(arr)
            h, w = arr.shape[:2]
            c = 1 if arr.ndim == 2 else arr.shape[2]
            fr.tex[slot] = Texture(w, h, c, 0, arr.ctypes.data)
        if cubemap is not None:
            faces = [None if f is None else np.ascontiguousarray(f, dtype=np.uint8) for f in cubemap]
            self._keep.append(faces)
            first = next(f for f in faces if f is not None)
            fr.skybox.face_size = first.shape[0]
            fr.skybox.channels = first.shape[2]
            for i, f in enumerate(faces):
                fr.skybox.faces[i] = None if f is None else f.ctypes.data
            fr.skybox.gen_mipmap = 1 if cube_mipmap else 0
        fr.texture_lod = texture_lod
        self.frame = fr
        self.width, self.height = fb_width, fb_height

    def mip_levels(self, uniform: str) -> list:
        """The RGBA8 mip levels >= 1 of the texture bound to sampler `uniform` as the oracle builds them ((h, w, 4) uint8 each)."""
        t = self.frame.tex[TEX_SLOTS.index(uniform)]
        l = lib()
        l.orc_kat_mip_level.restype = ctypes.c_int
        l.orc_kat_mip_level.argtypes = [ctypes.POINTER(Texture), ctypes.c_int, ctypes.c_void_p]
        out, L = [], 1
        while True:
            buf = np.empty(t.width * t.height * 4, np.uint8)
            r = l.orc_kat_mip_level(ctypes.byref(t), L, buf.ctypes.data)
            if r == 0:
                return out
            w, h = r >> 16, r & 0xffff
            out.append(np.ascontiguousarray(buf[: w * h * 4].reshape(h, w, 4)))
            L += 1

    def set_mip_levels(self, uniform: str, levels):
        """Diagnostic: replace the mip levels >= 1 of that texture by the caller's (orc_kat_set_mip_level); None / empty = leave.
        Call drop_mips() to return to the oracle's own."""
        t = self.frame.tex[TEX_SLOTS.index(uniform)]
        l = lib()
        l.orc_kat_set_mip_level.restype = ctypes.c_int
        l.orc_kat_set_mip_level.argtypes = [ctypes.POINTER(Texture), ctypes.c_int, ctypes.c_void_p]
        for L, lv in enumerate(levels or (), start=1):
            a = np.ascontiguousarray(lv, np.uint8)
            if l.orc_kat_set_mip_level(ctypes.byref(t), L, a.ctypes.data) != 0:
                raise RuntimeError(f"mip level {L} of {uniform} does not exist")

    @staticmethod
    def drop_mips():
        lib().orc_kat_drop_mips()

    def primary_hits(self, threads: int = 0, torus_t=None, y0: int = 0, y1: int | None = None):
        """Diagnostic (orc_set_primary_buffers): -> (frame, hits) where hits[y, x] = (t, type, num, 0) of the pixel's FIRST calcInter
        (type = num = -1 on a miss). torus_t: optional (H, W) float32 -- where the camera ray hits a torus and the entry is > 0, that
        distance replaces the solver's own root (the reference's root substituted)."""
        l = lib()
        l.orc_set_primary_buffers.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        hits = np.zeros((self.height, self.width, 4), np.float32)
        tin = None if torus_t is None else np.ascontiguousarray(torus_t, np.float32)
        assert tin is None or tin.shape == (self.height, self.width)
        l.orc_set_primary_buffers(hits.ctypes.data, None if tin is None else tin.ctypes.data)
        try:
            frame, _ = self.render(y0, y1, threads=threads)     # (a row range renders those rows only; the buffers stay whole-frame)
        finally:
            l.orc_set_primary_buffers(None, None)
        return frame, hits

    def render(self, y0: int = 0, y1: int | None = None, threads: int = 0, jitter=(0.0, 0.0), tags=None, lod_force: float = -1.0, lod_force_site=None):
        """Returns (float32 array (rows, W, 4), counters dict). Row 0 = bottom (gl_FragCoord).
        jitter: diagnostic displacement of every primary ray (orc_set_ray_jitter), in units of the un-normalised view vector.
        tags: optional (rows, W) uint32 array that receives the per-pixel ORC_TAG_* event bits (TAG_* below).
        lod_force: diagnostic, >= 0: every mip-mapped fetch is sampled at this level of detail (orc_set_lod_force).
        lod_force_site: diagnostic, up to two (sampler uniform, site 0 = hit / 1 = shadow, level): the fetches of those sites at levels of their own."""
        y1 = self.height if y1 is None else y1
        if threads <= 0:
            threads = effective_cpus()[0]      # "all cores" = the CPUs this container may use, not the hardware threads it can see
        out = np.empty((y1 - y0, self.width, 4), dtype=np.float32)
        cnt = Counters()
        l = lib()
        l.orc_set_ray_jitter.argtypes = [ctypes.c_float, ctypes.c_float]
        l.orc_set_tag_buffer.argtypes = [ctypes.c_void_p]
        l.orc_set_ray_jitter(float(jitter[0]), float(jitter[1]))
        l.orc_set_lod_force.argtypes = [ctypes.c_float]
        l.orc_set_lod_force(float(lod_force))
        l.orc_set_lod_force_site.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float]
        for idx, f in enumerate(lod_force_site or ()):
            l.orc_set_lod_force_site(idx, TEX_SLOTS.index(f[0]), int(f[1]), float(f[2]))
        if tags is not None:
            assert tags.shape == (y1 - y0, self.width) and tags.dtype == np.uint32 and tags.flags.c_contiguous
            l.orc_set_tag_buffer(tags.ctypes.data)
        # the level-0 texels are re-hashed (a stale mip chain under a reused address is dropped) on this object's FIRST render only: its arrays
        # are pinned in self._keep from then on, and the accounting of tests/reference_classify.py renders one description thousands of times
        l.orc_set_mip_validation.argtypes = [ctypes.c_int]
        l.orc_set_mip_validation(0 if getattr(self, "_validated", False) else 1)
        try:
            rc = l.orc_render(ctypes.byref(self.frame), y0, y1, out.ctypes.data, ctypes.byref(cnt), threads)
            self._validated = True
        finally:
            l.orc_set_mip_validation(1)
            l.orc_set_ray_jitter(0.0, 0.0)
            l.orc_set_lod_force(-1.0)
            l.orc_set_lod_force_site(0, -1, -1, -1.0)
            l.orc_set_lod_force_site(1, -1, -1, -1.0)
            l.orc_set_tag_buffer(None)
        if rc != 0:
            raise RuntimeError("orc_render failed")
        return out, cnt.as_dict()
