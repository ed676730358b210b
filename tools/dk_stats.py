#!/usr/bin/env python3
"""Diagnostic (GPU box): how densely populated are the waves that run the torus quartic solver?
Needs the -DRT_DK_STATS build of the library (RTX_HIP_LIB=.../librtx_hip_dk.so)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_opengl_amd import _capi, scenes, textures, wrapper  # noqa: E402

W, H = 3840, 2160


def main():
    lib = _capi.load()
    fn = lib.rtx_debug_dk_stats
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    out = (ctypes.c_ulonglong * 16)()
    ts = textures.default_texture_set()
    for name in sys.argv[1:] or ["default"]:
        name, _, depth = name.partition(':')
        sc = scenes.build_scene(name, W, H, int(depth or 4))
        gl = wrapper.make_renderer(sc, W, H, ts["textures"], ts["cubemap"])
        gl.draw(); gl.finish()
        fn(out, 1)
        gl.draw(); gl.finish()
        fn(out, 1)
        v = list(out)
        for _ in range(5):
            gl.draw()
        gl.finish()
        ms = gl.sum_recent_draw_ms(5) / 5
        print(f"{name}: kernel {ms:.3f} ms (instrumented)")
        print(f"  wave-level solver runs {v[0]}, lane-level solves {v[1]}, mean lanes/run {v[1]/max(v[0],1):.1f}")
        print(f"  wave sweeps {v[2]} (mean {v[2]/max(v[0],1):.1f}/run), lane sweeps {v[3]} (mean {v[3]/max(v[1],1):.1f}/solve)")
        print(f"  lane-sweep utilisation (lane sweeps / 64 x wave sweeps): {v[3] / max(64 * v[2], 1):.3f}")
        print(f"  wave cycles in solver {v[4]} (mean {v[4]/max(v[0],1):.0f}/run, {v[4]/max(v[2],1):.0f}/sweep)")
        print(f"  lane solves: accepted hit {v[7]}, real root beyond the limit {v[5]}, no usable root {v[1]-v[7]-v[5]}; hitting 60 sweeps {v[6]}")
        print(f"  torus scans with a candidate {v[8]}: passes (busiest lane's candidates) {v[9]} = {v[9]/max(v[8],1):.2f}/scan, candidates {v[10]} = "
              f"{v[10]/max(v[8],1):.1f}/scan over {v[12]/max(v[8],1):.1f} lanes; passes if idle lanes took the extra candidates: {v[11]} = {v[11]/max(v[8],1):.2f}/scan")
        print(f"  solver runs with a lane at the 60-sweep cap: {v[13]} ({v[6]/max(v[13],1):.1f} capped lanes each); without their capped lanes those runs would take {v[14]/max(v[13],1):.1f} sweeps")
        gl.stop()


if __name__ == "__main__":
    main()
