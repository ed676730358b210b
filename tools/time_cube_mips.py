#!/usr/bin/env python3
"""GPU box: what does a mip-mapped sky box (GLWrapper::load_cubemap(faces, true): the SKYLOD kernel instantiations) cost? The three bench
scenes at 4K (and the default scene at 1920x1080) with genMipmap off / on: best-of kernel time by the launch's own events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_opengl_amd import scenes, textures, wrapper  # noqa: E402


def main():
    ts = textures.default_texture_set()
    for name, depth, W, H in (("default", 4, 3840, 2160), ("default", 4, 1920, 1080), ("quadric", 4, 3840, 2160), ("torus", 6, 3840, 2160)):
        sc = scenes.build_scene(name, W, H, depth)
        row = []
        for mips in (False, True):
            gl = wrapper.make_renderer(sc, W, H, ts["textures"], ts["cubemap"], cube_mipmap=mips)
            for _ in range(5):
                gl.draw()
            gl.finish()
            best = 1e9
            for _ in range(3):
                for _ in range(20):
                    gl.draw()
                gl.finish()
                best = min(best, gl.sum_recent_draw_ms(20) / 20)
            row.append(best * 1000)
            gl.stop()
        print(f"{name} {W}x{H} d{depth}: genMipmap off {row[0]:8.1f} us | on {row[1]:8.1f} us ({100 * (row[1] / row[0] - 1):+.1f} %)", flush=True)


if __name__ == "__main__":
    main()
