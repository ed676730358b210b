#!/usr/bin/env python3
"""Cost of a scene update (the reference re-uploads its blocks every frame, main.cpp:167-169): host re-pack + upload + ray-pencil mask build
+ trace, per frame, against frames of a static scene.   usage: tools/time_scene_update.py [scene] [depth] [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracing_opengl_amd import scenes, textures, wrapper  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "quadric"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 30
W, H = 3840, 2160
ts = textures.default_texture_set(scale=4)
sc = scenes.build_scene(kind, W, H, depth)
gl = wrapper.make_renderer(sc, W, H, ts["textures"], ts["cubemap"])
for pencils in (1, 0):
    gl.set_option(wrapper.RTX_OPT_RAY_PENCILS, pencils)
    for mode in ("static", "updated every frame"):
        for _ in range(3):
            gl.draw()
        gl.finish()
        build = []
        t0 = time.perf_counter()
        for _ in range(frames):
            if mode != "static":
                gl.uploader.update()
            gl.draw()
            if mode != "static":
                gl.finish()
                build.append(gl.stats()["last_pencil_build_ms"])
        gl.finish()
        dt = (time.perf_counter() - t0) / frames * 1e3
        st = gl.stats()
        print(f"{kind} pencils={pencils} {mode:20s}: {dt:7.3f} ms/frame wall, trace kernel {st['last_draw_ms']:.3f} ms"
              + (f", pencil build {sorted(build)[len(build) // 2]:.3f} ms (median), {st['pencils']} pencils" if build and pencils else ""))
gl.stop()
