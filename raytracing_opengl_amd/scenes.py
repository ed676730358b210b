"""Scene blocks for the tracer: the exact uniform-block bytes SceneManager uploads.

Scenes are built by the C++ host library (librtx_host.so: this repo's SceneManager /
SurfaceFactory / scene.h headers, reference src/SceneManager.cpp:137-276, src/Surface.h,
src/main.cpp:43-132,197-246) and come back as an RTXB container (csrc/host/scene_blob.h).
"""
from __future__ import annotations

import ctypes
import os
import struct
from dataclasses import dataclass, field

_HERE = os.path.dirname(os.path.abspath(__file__))

# binding-point order of SceneManager::init_buffers (reference SceneManager.cpp:244-255)
BLOCK_NAMES = ("scene_buf", "spheres_buf", "planes_buf", "surfaces_buf", "boxes_buf", "toruses_buf",
               "rings_buf", "lights_point_buf", "lights_direct_buf")
DEFINES_FMT = "<9i6f"  # rt_defines, reference src/scene.h:7-20
RTXB_MAGIC = 0x42585452


@dataclass
class SceneBlocks:
    """One frame's worth of tracer inputs on the host: defines + nine named blocks."""
    defines: tuple            # (sphere,plane,surface,box,torus,ring,light_point,light_direct,iterations, amb r,g,b, shadow r,g,b)
    blocks: dict = field(default_factory=dict)  # name -> bytes

    @property
    def canvas(self):
        w, h = struct.unpack_from("<2i", self.blocks["scene_buf"], 44)
        return w, h

    def defines_bytes(self) -> bytes:
        return struct.pack(DEFINES_FMT, *self.defines)


def parse_rtxb(blob: bytes) -> SceneBlocks:
    magic, *sizes = struct.unpack_from("<10I", blob, 0)
    if magic != RTXB_MAGIC:
        raise ValueError("not an RTXB container")
    off = 40
    defines = struct.unpack_from(DEFINES_FMT, blob, off)
    off += struct.calcsize(DEFINES_FMT)
    blocks = {}
    for name, size in zip(BLOCK_NAMES, sizes):
        blocks[name] = bytes(blob[off:off + size])
        off += size
    if off != len(blob):
        raise ValueError("RTXB container length mismatch")
    return SceneBlocks(defines=defines, blocks=blocks)


_host = None


def _host_lib():
    global _host
    if _host is None:
        path = os.path.join(_HERE, "librtx_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (or make -C raytracing_opengl_amd)")
        lib = ctypes.CDLL(path)
        lib.rtxh_scene_build.restype = ctypes.c_size_t
        lib.rtxh_scene_build.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                         ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        _host = lib
    return _host


def build_scene(kind: str, width: int, height: int, depth: int, time: float = 0.0, delta: float = 0.0,
                yaw: float = 0.0, pitch: float = 0.0, cam_pos=None) -> SceneBlocks:
    """kind: 'default' | 'quadric' | 'torus' (BASELINE.json configs; recipes in csrc/host/scene_recipes.h).

    width/height are the scene canvas; like reference main.cpp:39-41 odd sizes are bumped to even.
    """
    if width % 2 == 1:
        width += 1
    if height % 2 == 1:
        height += 1
    lib = _host_lib()
    cam = (ctypes.c_float * 3)(*cam_pos) if cam_pos is not None else None
    need = lib.rtxh_scene_build(kind.encode(), width, height, depth, time, delta, yaw, pitch, cam, None, 0)
    if need == 0:
        raise ValueError(f"unknown scene kind {kind!r}")
    buf = ctypes.create_string_buffer(need)
    lib.rtxh_scene_build(kind.encode(), width, height, depth, time, delta, yaw, pitch, cam, buf, need)
    return parse_rtxb(buf.raw)
