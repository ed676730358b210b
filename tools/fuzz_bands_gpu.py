#!/usr/bin/env python3
"""GPU box: random frame sizes / rank counts / band heights: the bands traced by every 'rank' (rtx_draw_bands), un-permuted
(bands.unpermute), must equal the full frame bit for bit, for both colour targets."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from raytracing_opengl_amd import bands, scenes, textures, wrapper  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(7)
    ts = textures.default_texture_set(scale=16)
    bad = 0
    for k in range(n):
        w, h = int(rng.integers(17, 400)), int(rng.integers(9, 300))
        world = int(rng.integers(1, 9))
        band_rows = 8 * int(rng.integers(1, 5))
        name = ["default", "quadric", "torus"][k % 3]
        sc = scenes.build_scene(name, w, h, int(rng.integers(1, 5)))
        gl = wrapper.make_renderer(sc, w, h, ts["textures"], ts["cubemap"])
        gl.draw()
        for fmt, dtype in ((wrapper.RTX_RGBA32F, torch.float32), (wrapper.RTX_RGBA8, torch.uint8)):
            full = torch.from_numpy(gl.read_pixels(fmt).copy()).view(h, w, 4)
            rows_max = bands.max_local_rows(h, band_rows, world)
            parts = []
            for r in range(world):
                buf = torch.zeros((rows_max, w, 4), dtype=dtype, device="cuda:0")
                gl.draw_bands(band_rows, r, world, buf.data_ptr(), fmt)
                gl.finish()
                parts.append(buf)
            frame = bands.unpermute(torch.stack(parts), h, band_rows, world).cpu()
            same = torch.equal(frame.view(torch.uint8), full.view(torch.uint8)) if dtype == torch.uint8 else torch.equal(frame.view(torch.int32), full.view(torch.int32))
            if not same:
                bad += 1
                print(f"MISMATCH {name} {w}x{h} world {world} band_rows {band_rows} fmt {fmt}", flush=True)
        gl.stop()
    print(f"{n} random band configurations x 2 targets: {bad} mismatches")


if __name__ == "__main__":
    main()
