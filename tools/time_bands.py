#!/usr/bin/env python3
"""GPU box: kernel time of ONE rank's share of the default frame (4K, or `time_bands.py W H`) for N = 1, 2, 4, 8 ranks
(interleaved 8-row bands, RGBA8 target) -- what each GPU of a multi-GPU run traces, measured on one GPU. Shows how far
per-rank work is from 1/N."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from raytracing_opengl_amd import bands, scenes, textures, wrapper  # noqa: E402


def main():
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
    sc = scenes.build_scene("default", w, h, 4)
    if os.environ.get("NO_TORUS"):   # ablation: is the per-rank floor the torus solver's long waves?
        d = list(sc.defines); d[4] = 0
        sc.defines = tuple(d); sc.blocks["toruses_buf"] = b""
    ts = textures.default_texture_set()
    gl = wrapper.make_renderer(sc, w, h, ts["textures"], ts["cubemap"])
    stream = torch.cuda.current_stream().cuda_stream
    for world in (1, 2, 4, 8):
        rows_max = bands.max_local_rows(h, 8, world)
        buf = torch.empty((rows_max, w, 4), dtype=torch.uint8, device="cuda:0")
        times = []
        for rank in range(world):
            for _ in range(5):
                gl.draw_bands(8, rank, world, buf.data_ptr(), wrapper.RTX_RGBA8, stream)
            gl.finish(); gl.stats()
            best = 1e9
            for _ in range(3):
                for _ in range(20):
                    gl.draw_bands(8, rank, world, buf.data_ptr(), wrapper.RTX_RGBA8, stream)
                gl.finish()
                best = min(best, gl.sum_recent_draw_ms(20) / 20)
            times.append(best * 1000)
        if world == 1:
            base = times[0]
        print(f"N={world}: per-rank kernel us min {min(times):.1f} max {max(times):.1f}  (1/N of the one-GPU frame would be {base/world:.1f})", flush=True)
    gl.stop()


if __name__ == "__main__":
    main()
