#!/usr/bin/env python3
"""GPU box: kernel time of the default scene at several animation times (the reference's main loop updates the scene every frame:
main.cpp:197-246; at t = 0 the crate and the torus still have the rotations they were created with). 3840x2160, depth 4 unless given.
usage: tools/time_animated.py [width height depth]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from raytracing_opengl_amd import scenes, textures, wrapper  # noqa: E402


def main():
    W, H, D = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (3840, 2160, 4)
    ts = textures.default_texture_set()
    for t in (0.0, 1.0, 12.5, 40.0):
        sc = scenes.build_scene("default", W, H, D, time=t, delta=t)
        gl = wrapper.make_renderer(sc, W, H, ts["textures"], ts["cubemap"])
        for _ in range(3):
            gl.draw()
        gl.finish()
        best = 1e9
        for _ in range(3):
            for _ in range(20):
                gl.draw()
            gl.finish()
            best = min(best, gl.sum_recent_draw_ms(20) / 20)
        frame = gl.read_pixels(wrapper.RTX_RGBA32F)
        print(f"default scene {W}x{H} d{D} t = {t:5.1f}: kernel {best * 1000:7.1f} us  frame {hashlib.sha1(np.ascontiguousarray(frame).tobytes()).hexdigest()[:10]}", flush=True)
        gl.stop()


if __name__ == "__main__":
    main()
