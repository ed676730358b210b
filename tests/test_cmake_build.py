"""The CMake build (the reference's build system, /root/reference/CMakeLists.txt:73-89; this repo's CMakeLists.txt is what a maintainer
would add_subdirectory()) produces a loadable librtx_hip.so with every kernel in it, from the same sources, flags and occupancy targets as
the Makefile that __graft_entry__.build() drives (both read raytracing_opengl_amd/kernel_build.cfg).

Round 3's CMake list had lost smaa_kernel.hip and bands_kernel.hip (undefined smaa_launch / bands_unpack at link time) and compiled the
many-primitive variant for 7 waves per SIMD while the Makefile shipped 6 -- nothing built through CMake, so nothing noticed."""
import ctypes
import os
import re
import shutil
import subprocess

import pytest

from raytracing_opengl_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _cfg():
    out = {}
    for line in open(os.path.join(ROOT, "raytracing_opengl_amd", "kernel_build.cfg")):
        m = re.match(r"^([A-Z_]+)\s*[?:]?=\s*(.*)$", line)
        if m:
            out[m.group(1)] = m.group(2).strip()
    return out


@pytest.fixture(scope="module")
def cmake_build(tmp_path_factory):
    if shutil.which("cmake") is None or not os.path.exists(CLANG):
        pytest.skip("no cmake / ROCm clang here")
    bdir = tmp_path_factory.mktemp("cmake_build")
    subprocess.run(["cmake", "-S", ROOT, "-B", str(bdir), f"-DCMAKE_HIP_COMPILER={CLANG}", "-DCMAKE_EXPORT_COMPILE_COMMANDS=ON"],
                   check=True, capture_output=True, text=True)
    r = subprocess.run(["cmake", "--build", str(bdir), "-j", "8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
    return bdir


def test_cmake_builds_links_and_loads(cmake_build):
    lib = os.path.join(cmake_build, "librtx_hip.so")
    assert os.path.exists(lib) and os.path.exists(os.path.join(cmake_build, "librtx_host.so"))
    assert os.path.exists(os.path.join(cmake_build, "demo_main")), "the main.cpp-style demo did not link against rtx_hip"
    # nothing of the SMAA / band-placement kernels may be left for the dynamic linker to find
    undefined = subprocess.run(["nm", "-D", "--undefined-only", lib], check=True, capture_output=True, text=True).stdout
    leftover = [ln for ln in undefined.splitlines() if re.search(r"\b(smaa_|bands_|rt_launch)", ln)]
    assert not leftover, leftover
    # RTLD_NOW: every symbol resolves at load time; every entry point of include/rtx.h is exported
    h = ctypes.CDLL(lib, mode=os.RTLD_NOW | os.RTLD_LOCAL)
    for name in _capi.SYMBOLS:
        assert hasattr(h, name), f"the CMake-built librtx_hip.so does not export {name}"
    assert b"rtx-hip" in ctypes.cast(h.rtx_version, ctypes.CFUNCTYPE(ctypes.c_char_p))()


def test_cmake_and_makefile_compile_the_same_configuration(cmake_build):
    """Same sources, flags and occupancy targets as the Makefile: both come from kernel_build.cfg."""
    import json
    cfg = _cfg()
    cmds = json.load(open(os.path.join(cmake_build, "compile_commands.json")))
    by_file = {os.path.basename(c["file"]): c["command"] for c in cmds}
    for src in cfg["KERNEL_SOURCES"].split():
        assert src in by_file, f"{src} is not part of the CMake build"
        cmd = by_file[src]
        assert f"-DRT_WAVES_PER_EU={cfg['WAVES_PER_EU']}" in cmd and f"-DRT_WPE_HEAVY={cfg['WPE_HEAVY']}" in cmd, cmd
        assert "--offload-arch=gfx950" in cmd, cmd
        flags = cfg["KERNEL_FLAGS"].replace("-mllvm -disable-machine-licm", "-disable-machine-licm").split()
        for flag in flags:
            assert flag in cmd or flag.replace("-std=c++17", "-std=gnu++17") in cmd, (flag, cmd)
    mk = open(os.path.join(ROOT, "raytracing_opengl_amd", "Makefile")).read()
    assert "include $(HERE)kernel_build.cfg" in mk and "$(KERNEL_FLAGS)" in mk and "$(KERNEL_SOURCES)" in mk
    assert not re.search(r"^WAVES_PER_EU\s*[?:]?=", mk, re.M), "the Makefile must take the occupancy targets from kernel_build.cfg"
