#!/usr/bin/env python3
"""GPU box: frames back to back on ONE stream (what bench.py times) against the same frames alternating over TWO streams,
so that the next frame's workgroups fill the slots the previous frame's tail leaves idle. 4K default scene, depth 4."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from raytracing_opengl_amd import scenes, textures, wrapper  # noqa: E402


def main():
    w, h = 3840, 2160
    sc = scenes.build_scene("default", w, h, 4)
    ts = textures.default_texture_set()
    gl = wrapper.make_renderer(sc, w, h, ts["textures"], ts["cubemap"])
    band_rows = ((h + 7) // 8) * 8
    bufs = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda:0") for _ in range(3)]
    n = 400
    for n_streams in (1, 2, 3):
        streams = [torch.cuda.Stream() for _ in range(n_streams)]
        for k in range(10):
            gl.draw_bands(band_rows, 0, 1, bufs[k % n_streams].data_ptr(), wrapper.RTX_RGBA32F, streams[k % n_streams].cuda_stream)
        torch.cuda.synchronize()
        gl.stats()
        best = 1e9
        for _rep in range(3):
            t0 = time.perf_counter()
            for k in range(n):
                gl.draw_bands(band_rows, 0, 1, bufs[k % n_streams].data_ptr(), wrapper.RTX_RGBA32F, streams[k % n_streams].cuda_stream)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / n)
            gl.stats()
        print(f"{n_streams} stream(s): {best*1e3:.4f} ms/frame")
    gl.stop()


if __name__ == "__main__":
    main()
