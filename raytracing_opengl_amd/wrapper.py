"""Python host-side mirror of the reference's GLWrapper / SceneManager upload path.

Same method names, argument meaning and call order as reference src/GLWrapper.h:17-38 and
src/SceneManager.cpp:238-276, bound to the C ABI (include/rtx.h) with ctypes.  Image-file
decoding is outside the replaced path: textures are handed over as uint8 arrays.
Errors raise RtxError (the reference prints and exit()s).
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _capi
from ._capi import (RTX_OPT_BAND_LAYOUT, RTX_OPT_GATHER_RGB, RTX_GATHER_PEER_COPY, RTX_GATHER_RCCL, RTX_GATHER_RCCL_LOOPBACK, RTX_OPT_GATHER_TARGETS, RTX_OPT_COUNT_RAYS, RTX_OPT_CULL, RTX_OPT_HIGH_OCCUPANCY, RTX_OPT_HOT_ROWS_FIRST, RTX_OPT_RAY_PENCILS, RTX_OPT_SCENE_LDS, RTX_OPT_TEXTURE_LOD, RTX_OPT_XCD_REMAP, RTX_RGBA8, RTX_RGBA32F,
                    RTX_SCREEN_RGBA8, RTX_SMAA_EDGES_RG8, RTX_SMAA_HIGH, RTX_SMAA_LOW, RTX_SMAA_MEDIUM, RTX_SMAA_OFF, RTX_SMAA_ULTRA, RTX_SMAA_WEIGHTS_RGBA8,
                    RTX_WRAP_CLAMP_TO_EDGE, RTX_WRAP_REPEAT)

LOW, MEDIUM, HIGH, ULTRA = RTX_SMAA_LOW, RTX_SMAA_MEDIUM, RTX_SMAA_HIGH, RTX_SMAA_ULTRA   # enum SMAA_PRESET (reference src/SMAA_Builder.h:9-12)


class RtxError(RuntimeError):
    pass


def _check(status: int, what: str):
    if status != _capi.RTX_OK:
        raise RtxError(f"{what}: {_capi.load().rtx_last_error().decode()} (status {status})")


class GLWrapper:
    """GLWrapper(width, height, fullScreen) -- reference src/GLWrapper.cpp:12-18."""

    def __init__(self, width: int, height: int, fullScreen: bool = False, device: int = 0, devices=None, gather: int = _capi.RTX_GATHER_RCCL,
                 rank=None):
        """devices: list of HIP device ids -> a multi-device context (rtx_create_multi): the frame is split into interleaved row bands over
        them and assembled on devices[0]; gather: RTX_GATHER_RCCL, RTX_GATHER_PEER_COPY or RTX_GATHER_RCCL_LOOPBACK.
        rank: (rank, n_ranks, unique_id bytes) -> ONE rank of a frame split over n_ranks processes, on `device` (rtx_create_rank; the id
        comes from rccl_unique_id() on rank 0, handed round by the launcher). Default: the single device `device`."""
        self.width, self.height = int(width), int(height)
        self.device = int(device)
        self.devices = None if devices is None else [int(d) for d in devices]
        self.gather = int(gather)
        self.rank_spec = rank
        self._ctx = None
        self._lib = _capi.load()
        self.window = None  # no GLFW window: presentation is outside the replaced path

    # --- lifetime -------------------------------------------------------------------------
    def init_window(self) -> bool:
        """Creates the device context + colour target; False on failure (GLWrapper.cpp:61-133)."""
        ctx = ctypes.c_void_p()
        if self.rank_spec is not None:
            rank, n_ranks, uid = self.rank_spec
            assert len(uid) == _capi.RTX_RCCL_ID_BYTES
            buf = ctypes.create_string_buffer(bytes(uid), _capi.RTX_RCCL_ID_BYTES)
            status = self._lib.rtx_create_rank(self.width, self.height, self.device, int(rank), int(n_ranks), buf, self.gather, ctypes.byref(ctx))
        elif self.devices is not None:
            ids = (ctypes.c_int * len(self.devices))(*self.devices)
            status = self._lib.rtx_create_multi(self.width, self.height, len(self.devices), ids, self.gather, ctypes.byref(ctx))
        else:
            status = self._lib.rtx_create(self.width, self.height, self.device, ctypes.byref(ctx))
        if status != _capi.RTX_OK:
            self.last_error = self._lib.rtx_last_error().decode()
            return False
        self._ctx = ctx
        if getattr(self, "_smaa", None) is not None:
            _check(self._lib.rtx_enable_smaa(self._ctx, self._smaa), "enable_SMAA")
        return True

    def stop(self):
        if self._ctx is not None:
            self._lib.rtx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.stop()
        except Exception:
            pass

    def getWidth(self):
        return self.width

    def getHeight(self):
        return self.height

    def enable_SMAA(self, preset=ULTRA):
        """GLWrapper::enable_SMAA (GLWrapper.cpp:149-153): every draw() is followed by the three SMAA passes; read the result with
        read_pixels(RTX_SCREEN_RGBA8). Before init_window (the reference's order) the choice is remembered and applied there."""
        if isinstance(preset, str):
            preset = ("LOW", "MEDIUM", "HIGH", "ULTRA").index(preset)
        self._smaa = int(preset)
        if self._ctx is not None:
            _check(self._lib.rtx_enable_smaa(self._ctx, self._smaa), "enable_SMAA")

    def set_smaa_tables(self, area: np.ndarray, search: np.ndarray):
        """SMAA_Builder::load_area_texture / load_search_texture (SMAA_Builder.h:52-83): area (560, 160, 2) uint8, search (16, 64) uint8."""
        area = np.ascontiguousarray(area, np.uint8)
        search = np.ascontiguousarray(search, np.uint8)
        _check(self._lib.rtx_smaa_set_tables(self._ctx, area.ctypes.data, area.shape[1], area.shape[0], search.ctypes.data, search.shape[1], search.shape[0]),
               "set_smaa_tables")

    def smaa_resolve(self):
        """The post-process alone on the current RGBA8 colour target (GLWrapper.cpp:173-204)."""
        _check(self._lib.rtx_smaa_resolve(self._ctx), "smaa_resolve")

    def write_pixels(self, rgba8: np.ndarray):
        """Replace the RGBA8 colour target: (H, W, 4) uint8, row 0 = bottom row."""
        a = np.ascontiguousarray(rgba8, np.uint8)
        assert a.shape == (self.height, self.width, 4), a.shape
        _check(self._lib.rtx_write_pixels(self._ctx, RTX_RGBA8, a.ctypes.data, a.nbytes), "write_pixels")

    # --- specialisation / blocks ----------------------------------------------------------
    def init_shaders(self, defines):
        """defines: 15-tuple in rt_defines order or a _capi.Defines (GLWrapper.cpp:232-277)."""
        if not isinstance(defines, _capi.Defines):
            d = _capi.Defines()
            for k, (name, _t) in enumerate(_capi.Defines._fields_[:9]):
                setattr(d, name, int(defines[k]))
            d.ambient_color = (ctypes.c_float * 3)(*defines[9:12])
            d.shadow_ambient = (ctypes.c_float * 3)(*defines[12:15])
            defines = d
        _check(self._lib.rtx_specialize(self._ctx, ctypes.byref(defines)), "init_shaders")

    def init_buffer(self, name: str, bindingPoint: int, data: bytes | None, size: int | None = None) -> int:
        """Returns the block handle (the reference writes it through GLuint* ubo; GLWrapper.cpp:365-379)."""
        size = len(data) if (size is None and data is not None) else int(size or 0)
        handle = ctypes.c_uint32()
        buf = ctypes.create_string_buffer(data, size) if data else None
        _check(self._lib.rtx_block_create(self._ctx, name.encode(), bindingPoint, size, buf, ctypes.byref(handle)), "init_buffer")
        return handle.value

    def update_buffer(self, ubo: int, data: bytes):
        """GLWrapper::update_buffer (static in the reference; GLWrapper.cpp:381-386)."""
        buf = ctypes.create_string_buffer(data, len(data))
        _check(self._lib.rtx_block_update(self._ctx, ubo, len(data), buf), "update_buffer")

    # --- textures -------------------------------------------------------------------------
    def load_cubemap(self, faces, genMipmap: bool = False) -> int:
        """faces: six HxWxC uint8 arrays (+X,-X,+Y,-Y,+Z,-Z) or None entries (GLWrapper.cpp:284-317)."""
        arrs = [None if f is None else np.ascontiguousarray(f, dtype=np.uint8) for f in faces]
        first = next((a for a in arrs if a is not None), None)
        ptrs = (ctypes.c_void_p * 6)(*[None if a is None else a.ctypes.data for a in arrs])
        handle = ctypes.c_uint32()
        size = 0 if first is None else first.shape[0]
        ch = 3 if first is None else first.shape[2]
        _check(self._lib.rtx_cubemap_create(self._ctx, size, ch, ptrs, 1 if genMipmap else 0, ctypes.byref(handle)), "load_cubemap")
        return handle.value

    def set_skybox(self, textureId: int):
        """GLWrapper.cpp:135-141: sampler 'skybox' -> unit 0, bind the cubemap there."""
        _check(self._lib.rtx_sampler_unit(self._ctx, b"skybox", 0), "set_skybox")
        _check(self._lib.rtx_bind_texture(self._ctx, 0, textureId), "set_skybox")

    def load_texture(self, texNum: int, image, uniformName: str, wrapMode: int = RTX_WRAP_REPEAT) -> int:
        """image: HxWxC (or HxW) uint8 array, row 0 = t 0 (GLWrapper.cpp:319-363)."""
        arr = np.ascontiguousarray(image, dtype=np.uint8)
        h, w = arr.shape[:2]
        c = 1 if arr.ndim == 2 else arr.shape[2]
        handle = ctypes.c_uint32()
        _check(self._lib.rtx_texture2d_create(self._ctx, w, h, c, arr.ctypes.data, wrapMode, ctypes.byref(handle)), "load_texture")
        _check(self._lib.rtx_sampler_unit(self._ctx, uniformName.encode(), texNum), "load_texture")
        _check(self._lib.rtx_bind_texture(self._ctx, texNum, handle.value), "load_texture")
        return handle.value

    def bind_texture(self, unit: int, handle: int):
        """glActiveTexture(GL_TEXTURE0+unit); glBindTexture(...)  (main.cpp:178-187)."""
        _check(self._lib.rtx_bind_texture(self._ctx, unit, handle), "bind_texture")

    # --- draw / read back -----------------------------------------------------------------
    def set_option(self, option: int, value: int):
        _check(self._lib.rtx_set_option(self._ctx, option, value), "set_option")

    def get_option(self, option: int) -> int:
        v = ctypes.c_int(0)
        _check(self._lib.rtx_get_option(self._ctx, option, ctypes.byref(v)), "get_option")
        return v.value

    def draw(self):
        """GLWrapper::draw (GLWrapper.cpp:155-165), asynchronous on the context's stream."""
        _check(self._lib.rtx_draw(self._ctx), "draw")

    def draw_bands(self, band_rows: int, band_first: int, band_stride: int, dst_device_ptr: int, fmt: int = RTX_RGBA32F, stream: int = 0):
        _check(self._lib.rtx_draw_bands(self._ctx, band_rows, band_first, band_stride, ctypes.c_void_p(dst_device_ptr), fmt,
                                        ctypes.c_void_p(stream) if stream else None), "draw_bands")

    def draw_rows(self, row_first: int, n_rows: int, dst_device_ptr: int, fmt: int = RTX_RGBA32F, stream: int = 0):
        _check(self._lib.rtx_draw_rows(self._ctx, row_first, n_rows, ctypes.c_void_p(dst_device_ptr), fmt, ctypes.c_void_p(stream) if stream else None), "draw_rows")

    def finish(self):
        _check(self._lib.rtx_finish(self._ctx), "finish")

    def read_pixels(self, fmt: int = RTX_RGBA32F) -> np.ndarray:
        """(H, W, 4) float32 or uint8; row 0 = bottom row (gl_FragCoord origin)."""
        out = np.empty((self.height, self.width, 2 if fmt == RTX_SMAA_EDGES_RG8 else 4), dtype=np.float32 if fmt == RTX_RGBA32F else np.uint8)
        _check(self._lib.rtx_read_pixels(self._ctx, fmt, out.ctypes.data, out.nbytes), "read_pixels")
        return out

    def stats(self) -> dict:
        s = _capi.Stats()
        _check(self._lib.rtx_get_stats(self._ctx, ctypes.byref(s)), "stats")
        return {n: getattr(s, n) for n, _t in _capi.Stats._fields_}

    # --- multi-device: how the frame is split (rtx.h RTX_OPT_BAND_LAYOUT) ---------------------------
    def n_ranks(self) -> int:
        n = ctypes.c_int()
        _check(self._lib.rtx_device_count(self._ctx, ctypes.byref(n)), "device_count")
        return n.value

    def set_band_split(self, rows_per_rank):
        arr = (ctypes.c_int * len(rows_per_rank))(*[int(v) for v in rows_per_rank])
        _check(self._lib.rtx_set_band_split(self._ctx, arr, len(rows_per_rank)), "set_band_split")

    def band_split(self):
        n = self.n_ranks()
        arr = (ctypes.c_int * n)()
        _check(self._lib.rtx_get_band_split(self._ctx, arr, n), "get_band_split")
        return list(arr)

    def rank_draw_ms(self):
        n = self.n_ranks()
        arr = (ctypes.c_float * n)()
        _check(self._lib.rtx_get_rank_draw_ms(self._ctx, arr, n), "get_rank_draw_ms")
        return list(arr)

    def recent_draw_ms(self, n: int):
        """HIP-event durations (ms) of the n most recent draws, most recent first; does not retire them."""
        arr = (ctypes.c_float * n)()
        _check(self._lib.rtx_recent_draw_ms(self._ctx, n, arr), "recent_draw_ms")
        return [float(v) for v in arr]

    def sum_recent_draw_ms(self, n: int) -> float:
        v = ctypes.c_float()
        _check(self._lib.rtx_sum_recent_draw_ms(self._ctx, n, ctypes.byref(v)), "sum_recent_draw_ms")
        return v.value

    def selftest(self) -> int:
        v = ctypes.c_int()
        _check(self._lib.rtx_selftest(self._ctx, ctypes.byref(v)), "selftest")
        return v.value


class SceneUploader:
    """SceneManager's upload plumbing (reference SceneManager.cpp:238-276) for a SceneBlocks object."""

    def __init__(self, scene_blocks, wrapper: GLWrapper):
        self.scene, self.wrapper, self.ubos = scene_blocks, wrapper, {}

    def init(self):
        from .scenes import BLOCK_NAMES
        for binding, name in enumerate(BLOCK_NAMES):
            data = self.scene.blocks.get(name, b"")
            if name == "scene_buf":  # the reference allocates scene_buf with NULL data and fills it on the first update
                self.ubos[name] = self.wrapper.init_buffer(name, binding, None, size=len(data))
            else:
                self.ubos[name] = self.wrapper.init_buffer(name, binding, data)
        self.update()

    def update(self, scene_blocks=None):
        """update_buffers(): every block except lights_direct (trap T19), skipping empty vectors."""
        if scene_blocks is not None:
            self.scene = scene_blocks
        for name, ubo in self.ubos.items():
            data = self.scene.blocks.get(name, b"")
            if name == "lights_direct_buf" or not data:
                continue
            self.wrapper.update_buffer(ubo, data)


def rccl_unique_id() -> bytes:
    """rtx_rccl_unique_id: the 128 bytes rank 0 creates and every process of a per-process frame split passes to GLWrapper(rank=...)."""
    buf = ctypes.create_string_buffer(_capi.RTX_RCCL_ID_BYTES)
    _check(_capi.load().rtx_rccl_unique_id(buf), "rccl_unique_id")
    return buf.raw


def make_renderer(scene_blocks, fb_width: int, fb_height: int, textures=None, cubemap=None, device: int = 0, texture_lod: int = 1, devices=None,
                  gather: int = _capi.RTX_GATHER_RCCL, rank=None, cube_mipmap: bool = False) -> GLWrapper:
    """The start-up sequence of reference main.cpp:25-157 for a prepared scene: context, specialise,
    skybox, textures, blocks. cube_mipmap: load_cubemap(faces, genMipmap = true) (main.cpp:137 passes the default, false)."""
    gl = GLWrapper(fb_width, fb_height, False, device=device, devices=devices, gather=gather, rank=rank)
    if not gl.init_window():
        raise RtxError(f"init_window failed: {getattr(gl, 'last_error', '')}")
    gl.init_shaders(scene_blocks.defines)
    if cubemap is not None:
        gl.set_skybox(gl.load_cubemap(cubemap, bool(cube_mipmap)))
    for uniform, unit, img in (textures or ()):
        gl.load_texture(unit, img, uniform)
    up = SceneUploader(scene_blocks, gl)
    up.init()
    gl.uploader = up
    gl.set_option(RTX_OPT_TEXTURE_LOD, texture_lod)
    return gl
