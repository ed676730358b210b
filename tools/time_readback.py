#!/usr/bin/env python3
"""GPU box: what the rate becomes when every frame is also copied to host memory (rtx_read_pixels) -- the PCIe-inclusive
figure DESIGN.md section 7 quotes next to the resident-frame `value` of bench.py. 4K default scene, depth 4."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from raytracing_opengl_amd import scenes, textures, wrapper  # noqa: E402


def main():
    w, h, depth = 3840, 2160, 4
    sc = scenes.build_scene("default", w, h, depth)
    ts = textures.default_texture_set()
    gl = wrapper.make_renderer(sc, w, h, ts["textures"], ts["cubemap"])
    gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    gl.draw(); gl.finish()
    st = gl.stats()
    rays = st["rays_closest"] + st["rays_shadow"]
    gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 0)
    n = 30
    for _ in range(5):
        gl.draw()
    gl.finish()
    t0 = time.perf_counter()
    for _ in range(n):
        gl.draw()
    gl.finish()
    dt = (time.perf_counter() - t0) / n
    print(f"draw only               : {dt*1e3:7.3f} ms/frame  {rays/dt/1e9:6.2f} Gray/s")
    for fmt, name, bpp in ((wrapper.RTX_RGBA32F, "RGBA32F", 16), (wrapper.RTX_RGBA8, "RGBA8", 4)):
        out = np.zeros((h, w, 4), dtype=np.float32 if bpp == 16 else np.uint8)   # pre-faulted, reused (pageable) destination

        def read():
            wrapper._check(gl._lib.rtx_read_pixels(gl._ctx, fmt, out.ctypes.data, out.nbytes), "read_pixels")
        gl.draw(); read()
        t0 = time.perf_counter()
        for _ in range(n):
            gl.draw()
            read()
        dt = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n):
            read()
        rd = (time.perf_counter() - t0) / n
        print(f"draw + read_pixels {name:7s}: {dt*1e3:7.3f} ms/frame  {rays/dt/1e9:6.2f} Gray/s   (read alone {rd*1e3:.3f} ms = {w*h*bpp/rd/1e9:.1f} GB/s into a reused pageable buffer)")
    gl.stop()


if __name__ == "__main__":
    main()
