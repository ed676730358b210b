#!/usr/bin/env python3
"""Per-phase wave-cycle breakdown of the trace kernel (GPU box; needs the -DRT_PHASE_TIMERS build:
RTX_HIP_LIB=raytracing_opengl_amd/variants/librtx_hip_prof.so)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_opengl_amd import _capi, scenes, textures, wrapper  # noqa: E402

NAMES = ["setup", "scan(closest)", "hit_info", "classify", "sky", "shade(total)", "DK(nested)", "apply", "shadow scans(nested in shade)", "trips", "  closest: planes+spheres(+ray setup)", "  closest: surfaces", "  closest: boxes", "  closest: tori", "  closest: rings", "  closest: lights", "wave total", "waves"]
scene = sys.argv[1] if len(sys.argv) > 1 else "default"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W, H = 3840, 2160
sc = scenes.build_scene(scene, W, H, depth)
ts = textures.default_texture_set(scale=1)
gl = wrapper.make_renderer(sc, W, H, ts["textures"], ts["cubemap"])
lib = _capi.load()
out = (ctypes.c_ulonglong * 18)()
for _ in range(3):
    gl.draw()
gl.finish()
lib.rtx_debug_phase_counters(out, 1)
N = 5
for _ in range(N):
    gl.draw()
gl.finish()
print("kernel ms (profiling build):", gl.sum_recent_draw_ms(N) / N)
lib.rtx_debug_phase_counters(out, 0)
total = out[16]
print(f"{scene} depth {depth}: waves/frame {out[17]//N}, mean wave lifetime {total/out[17]:.0f} cycles, trips/wave {out[9]/out[17]:.2f}")
import time
for k, name in enumerate(NAMES[:16]):
    if k == 9: continue
    print(f"  {name:34s} {out[k]/out[17]:10.0f} cycles/wave  {100.0*out[k]/total:5.1f} %")
gl.stop()
