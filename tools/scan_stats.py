#!/usr/bin/env python3
"""Diagnostic (GPU box): the quadric candidate lists a WAVE walks (the OR of its lanes' mask words) against the lists its lanes
would walk on their own.  Needs the -DRT_SCAN_STATS build (RTX_HIP_LIB=.../librtx_hip_scan.so)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_opengl_amd import _capi, scenes, textures, wrapper  # noqa: E402

W, H = 3840, 2160
KINDS = ["closest-hit, pencil (camera rays)", "closest-hit, slab tables (mirror / refracted rays)", "shadow, pencil", "shadow, slab tables"]


def main():
    lib = _capi.load()
    fn = lib.rtx_debug_scan_stats
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    out = (ctypes.c_ulonglong * 32)()
    ts = textures.default_texture_set()
    for name in sys.argv[1:] or ["quadric"]:
        depth = 6 if name == "torus" else 4
        sc = scenes.build_scene(name, W, H, depth)
        gl = wrapper.make_renderer(sc, W, H, ts["textures"], ts["cubemap"])
        gl.draw(); gl.finish()
        fn(out, 1)
        gl.draw(); gl.finish()
        fn(out, 1)
        v = list(out)
        print(f"{name} {W}x{H} depth {depth}: one frame")
        for k in range(4):
            r = v[8 * k: 8 * k + 8]
            n = max(r[0], 1)
            print(f"  {KINDS[k]}: {r[0]} word walks; per walk: wave OR {r[1]/n:.2f} bits, "
                  f"busiest lane {r[4]/n:.2f}, mean lane {r[2]/max(r[3],1):.2f} ({r[3]/n:.1f} lanes); "
                  f"second-level runs {r[5]/n:.2f} with {r[6]/max(r[5],1):.1f} lanes each, of which {r[7]/max(r[6],1):.2f} hit")
        gl.stop()


if __name__ == "__main__":
    main()
