"""The C-ABI library loads and exports every symbol include/rtx.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from raytracing_opengl_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_library_agree(built):
    header = open(os.path.join(ROOT, "include", "rtx.h")).read()
    declared = set(re.findall(r"RTX_API\s+[\w\s\*]+?\b(rtx_\w+)\s*\(", header))
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"librtx_hip.so does not export {name}"


def test_no_cpu_fallback(built):
    """Without a HIP device rtx_create must fail loudly (never fall back to a CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _capi.load()
    ctx = ctypes.c_void_p()
    status = lib.rtx_create(64, 64, 0, ctypes.byref(ctx))
    assert status == 2 and not ctx.value  # RTX_ERR_DEVICE
    assert b"no CPU fallback" in lib.rtx_last_error()


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under the package or include/ may reference it."""
    bad = []
    for base in ("raytracing_opengl_amd", "include"):
        for dirpath, _dirs, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".cpp", ".hip", ".c")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"liboracle|rt_oracle|smaa_oracle|from oracle|import oracle|orc_render|libharness", text):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_cubemap_mipmaps_are_built_not_dropped_in_the_source():
    """load_cubemap(faces, genMipmap=true) changes the reference's sky (GLWrapper.cpp:307-310: cube mips + a trilinear fetch). Rounds 1-3
    dropped the flag silently, rounds 4-5 refused it; now it is implemented (run-time checks: tests/test_gpu_cube_mips.py). Here: the
    implementation names its argument, builds a chain from it, hands it on to the other ranks unchanged, and the header says what it does."""
    src = open(os.path.join(ROOT, "raytracing_opengl_amd", "csrc", "rtx_capi.cpp")).read()
    body = src[src.index("int rtx_cubemap_create("):]
    body = body[:body.index("\n}\n")]
    assert "/*gen_mipmap*/" not in body and re.search(r"if \(gen_mipmap\) \{", body) and "build_mip_chain" in body
    assert "faces, gen_mipmap, handle)" in body and "faces, 0, handle)" not in body
    assert "gen_mipmap != 0 is" in open(os.path.join(ROOT, "include", "rtx.h")).read()
