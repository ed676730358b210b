"""ctypes prototypes of the C ABI (include/rtx.h) exported by librtx_hip.so.

The library holds the gfx950 kernels; there is no CPU fallback.  If the shared object is missing
the import fails loudly with the build hint instead of degrading to another path.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTX_HIP_LIB") or os.path.join(_HERE, "librtx_hip.so")  # RTX_HIP_LIB: A/B builds of the same library

RTX_OK = 0
RTX_RGBA32F, RTX_RGBA8, RTX_SCREEN_RGBA8, RTX_SMAA_EDGES_RG8, RTX_SMAA_WEIGHTS_RGBA8 = 0, 1, 2, 3, 4
RTX_SMAA_OFF, RTX_SMAA_LOW, RTX_SMAA_MEDIUM, RTX_SMAA_HIGH, RTX_SMAA_ULTRA = -1, 0, 1, 2, 3
RTX_WRAP_REPEAT, RTX_WRAP_CLAMP_TO_EDGE = 0, 1
RTX_OPT_CULL, RTX_OPT_COUNT_RAYS, RTX_OPT_SCENE_LDS, RTX_OPT_TEXTURE_LOD, RTX_OPT_XCD_REMAP, RTX_OPT_HIGH_OCCUPANCY, RTX_OPT_HOT_ROWS_FIRST, RTX_OPT_GATHER_TARGETS, RTX_OPT_RAY_PENCILS, RTX_OPT_BAND_LAYOUT, RTX_OPT_GATHER_RGB = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
RTX_GATHER_RCCL, RTX_GATHER_PEER_COPY, RTX_GATHER_RCCL_LOOPBACK = 0, 1, 2
RTX_RCCL_ID_BYTES = 128

# every symbol include/rtx.h declares (tests/test_capi_symbols.py checks the .so against this list)
SYMBOLS = (
    "rtx_last_error", "rtx_version", "rtx_create", "rtx_destroy", "rtx_current", "rtx_make_current", "rtx_get_size",
    "rtx_specialize", "rtx_block_create", "rtx_block_update", "rtx_texture2d_create", "rtx_cubemap_create",
    "rtx_sampler_unit", "rtx_bind_texture", "rtx_texture_destroy", "rtx_set_option", "rtx_get_option", "rtx_draw",
    "rtx_draw_bands", "rtx_draw_rows", "rtx_finish", "rtx_read_pixels", "rtx_framebuffer_device", "rtx_get_stats", "rtx_get_stats_sized",
    "rtx_sum_recent_draw_ms", "rtx_recent_draw_ms", "rtx_selftest",
    "rtx_enable_smaa", "rtx_smaa_set_tables", "rtx_smaa_default_tables", "rtx_smaa_resolve", "rtx_write_pixels",
    "rtx_create_multi", "rtx_device_count", "rtx_rank", "rtx_rccl_unique_id", "rtx_create_rank",
    "rtx_set_band_split", "rtx_get_band_split", "rtx_get_rank_draw_ms",
)


class Defines(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("sphere_size", "plane_size", "surface_size", "box_size", "torus_size", "ring_size",
                                              "light_point_size", "light_direct_size", "iterations")] + \
               [("ambient_color", ctypes.c_float * 3), ("shadow_ambient", ctypes.c_float * 3)]


class Stats(ctypes.Structure):
    _fields_ = [("last_draw_ms", ctypes.c_float), ("launches", ctypes.c_uint32), ("rays_closest", ctypes.c_uint64),
                ("rays_shadow", ctypes.c_uint64), ("rays_shadow_cast", ctypes.c_uint64), ("torus_solves", ctypes.c_uint64),
                ("last_smaa_ms", ctypes.c_float), ("last_gather_ms", ctypes.c_float), ("smaa_edge_pixels", ctypes.c_uint32),
                ("last_pencil_build_ms", ctypes.c_float), ("pencils", ctypes.c_uint32),
                ("kernel_variant", ctypes.c_uint32), ("candidate_tables", ctypes.c_uint32)]


_lib = None


def load():
    """dlopen librtx_hip.so and attach prototypes. Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP tracer is not built. Run `python -c \"import __graft_entry__ as g; g.build()\"` "
            "or `make -C raytracing_opengl_amd`. There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so, and the host layer
    # shares device memory and streams with torch (bench.py, bands.py). Let torch load its runtime
    # FIRST so that this library's libamdhip64.so.7 dependency binds to the same copy; loading
    # /opt/rocm's runtime first leaves torch unable to see the GPU.
    try:
        import torch  # noqa: F401
    except ImportError:  # pure-ctypes users without torch: /opt/rocm's runtime is used
        pass
    lib = ctypes.CDLL(LIB_PATH)
    c, P = ctypes, ctypes.POINTER
    vp, u32, i = c.c_void_p, c.c_uint32, c.c_int
    lib.rtx_last_error.restype = c.c_char_p
    lib.rtx_version.restype = c.c_char_p
    lib.rtx_create.argtypes = [i, i, i, P(vp)]
    lib.rtx_destroy.argtypes = [vp]
    lib.rtx_destroy.restype = None
    lib.rtx_current.restype = vp
    lib.rtx_make_current.argtypes = [vp]
    lib.rtx_get_size.argtypes = [vp, P(i), P(i)]
    lib.rtx_specialize.argtypes = [vp, P(Defines)]
    lib.rtx_block_create.argtypes = [vp, c.c_char_p, i, c.c_size_t, vp, P(u32)]
    lib.rtx_block_update.argtypes = [vp, u32, c.c_size_t, vp]
    lib.rtx_texture2d_create.argtypes = [vp, i, i, i, vp, i, P(u32)]
    lib.rtx_cubemap_create.argtypes = [vp, i, i, P(vp), i, P(u32)]
    lib.rtx_sampler_unit.argtypes = [vp, c.c_char_p, i]
    lib.rtx_bind_texture.argtypes = [vp, i, u32]
    lib.rtx_texture_destroy.argtypes = [vp, u32]
    lib.rtx_set_option.argtypes = [vp, i, i]
    lib.rtx_get_option.argtypes = [vp, i, P(i)]
    lib.rtx_draw.argtypes = [vp]
    lib.rtx_draw_bands.argtypes = [vp, i, i, i, vp, i, vp]
    lib.rtx_draw_rows.argtypes = [vp, i, i, vp, i, vp]
    lib.rtx_finish.argtypes = [vp]
    lib.rtx_read_pixels.argtypes = [vp, i, vp, c.c_size_t]
    lib.rtx_framebuffer_device.argtypes = [vp, i, P(vp)]
    lib.rtx_get_stats.argtypes = [vp, P(Stats)]
    lib.rtx_get_stats_sized.argtypes = [vp, vp, c.c_size_t]
    lib.rtx_sum_recent_draw_ms.argtypes = [vp, i, P(c.c_float)]
    lib.rtx_recent_draw_ms.argtypes = [vp, i, P(c.c_float)]
    lib.rtx_selftest.argtypes = [vp, P(i)]
    lib.rtx_create_multi.argtypes = [i, i, i, P(i), i, P(vp)]
    lib.rtx_device_count.argtypes = [vp, P(i)]
    lib.rtx_rank.argtypes = [vp, P(i)]
    lib.rtx_set_band_split.argtypes = [vp, P(i), i]
    lib.rtx_get_band_split.argtypes = [vp, P(i), i]
    lib.rtx_get_rank_draw_ms.argtypes = [vp, P(c.c_float), i]
    lib.rtx_rccl_unique_id.argtypes = [vp]
    lib.rtx_create_rank.argtypes = [i, i, i, i, i, vp, i, P(vp)]
    lib.rtx_enable_smaa.argtypes = [vp, i]
    lib.rtx_smaa_set_tables.argtypes = [vp, vp, i, i, vp, i, i]
    lib.rtx_smaa_resolve.argtypes = [vp]
    lib.rtx_smaa_default_tables.argtypes = [vp, c.c_size_t, vp, c.c_size_t]
    lib.rtx_write_pixels.argtypes = [vp, i, vp, c.c_size_t]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if fn.restype is c.c_int and name not in ("rtx_last_error", "rtx_version", "rtx_current", "rtx_destroy"):
            fn.restype = c.c_int
    _lib = lib
    return lib
