#!/usr/bin/env python3
"""GPU box: the SMAA kernels against the oracle, byte for byte, on seeded random frames -- synthetic patterns (tests/smaa_cases.pattern),
coarse noise, long straight and diagonal runs -- at random sizes (odd widths and heights included) and all four presets. The three textures
(edges, weights, screen) must be identical. usage: tools/fuzz_smaa_gpu.py first_seed count"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402

import smaa_cases  # noqa: E402
from oracle import smaa  # noqa: E402
from raytracing_opengl_amd import smaa_tables, wrapper  # noqa: E402


def frame(seed):
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(1, 700)), int(rng.integers(1, 300))
    kind = seed % 4
    if kind == 0:
        img = smaa_cases.pattern(seed, w, h)
    elif kind == 1:
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        img[..., :3] = (img[..., :3] // 128) * 128
    else:
        img = np.full((h, w, 4), 255, np.uint8)
        for _ in range(int(rng.integers(1, 12))):          # long horizontal / vertical / diagonal runs, some into the border
            c = rng.integers(0, 200, 3).astype(np.uint8)
            if rng.random() < 0.4:
                y = int(rng.integers(0, h)); x0 = int(rng.integers(-20, w)); n = int(rng.integers(1, 400))
                img[y, max(x0, 0):max(min(x0 + n, w), 0), :3] = c
            elif rng.random() < 0.66:
                x = int(rng.integers(0, w)); y0 = int(rng.integers(-20, h)); n = int(rng.integers(1, 300))
                img[max(y0, 0):max(min(y0 + n, h), 0), x, :3] = c
            else:
                x, y, n, s = int(rng.integers(0, w)), int(rng.integers(0, h)), int(rng.integers(1, 200)), int(rng.choice([-1, 1]))
                for k in range(n):
                    if 0 <= x + k < w and 0 <= y + s * k < h:
                        img[y + s * k, x + k:x + k + int(rng.integers(1, 4)), :3] = c
    img[..., 3] = 255
    return np.ascontiguousarray(img)


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    area, search = smaa_tables.area_table(), smaa_tables.search_table()
    bad = 0
    for seed in range(first, first + count):
        img = frame(seed)
        h, w = img.shape[:2]
        preset = smaa.PRESETS[(seed // 4) % 4]
        gl = wrapper.GLWrapper(w, h)
        gl.enable_SMAA(preset)
        assert gl.init_window(), getattr(gl, "last_error", "")
        for rep in range(2):                                 # twice: the clear-through-the-list invariant between frames
            gl.write_pixels(img if rep == 0 else img[::-1].copy())
            gl.smaa_resolve()
            got = {"edges": gl.read_pixels(wrapper.RTX_SMAA_EDGES_RG8), "blend": gl.read_pixels(wrapper.RTX_SMAA_WEIGHTS_RGBA8), "screen": gl.read_pixels(wrapper.RTX_SCREEN_RGBA8)}
            want = smaa.run(img if rep == 0 else img[::-1].copy(), preset, area, search)
            for k in ("edges", "blend", "screen"):
                if not np.array_equal(got[k], want[k]):
                    bad += 1
                    print(f"MISMATCH seed {seed} {w}x{h} {preset} rep {rep} {k}: {int((got[k] != want[k]).sum())} bytes", flush=True)
        gl.stop()
    print(f"smaa gpu fuzz: seeds {first}..{first + count - 1}: {bad} mismatching textures")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
