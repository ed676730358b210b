"""Inputs of the SMAA parity tests: RGBA8 frames (row 0 = bottom row) that exercise every branch of the three passes.

  traced frames   -- the tracer's own output (oracle render, quantised like the RGBA8 colour target) of the three bench scenes;
  pattern(seed)   -- synthetic: long axis-aligned edges running into the image border (search-length limit, clamp-to-edge),
                     lines at many slopes (diagonal detection both ways), circles, one-pixel features, corners, a checkerboard
                     (every pixel an edge pixel), low-contrast steps around each preset's threshold, noise."""
from __future__ import annotations

import numpy as np


def quantise(img32: np.ndarray) -> np.ndarray:
    v = np.clip(np.nan_to_num(img32.astype(np.float32), nan=0.0), 0.0, 1.0).astype(np.float32)
    return (v * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)


def traced(kind: str, w: int, h: int, depth: int, tex_scale: int = 16) -> np.ndarray:
    from oracle import oracle
    from raytracing_opengl_amd import scenes, textures
    ts = textures.default_texture_set(scale=tex_scale)
    sc = scenes.build_scene(kind, w, h, depth)
    img, _ = oracle.OracleScene(sc, w, h, ts["textures"], ts["cubemap"]).render()
    return quantise(img)


def pattern(seed: int, w: int, h: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w, 3), np.float64)
    img += np.array([0.10, 0.12, 0.15])
    img[h // 3:, :] = (0.55, 0.5, 0.45)                                   # full-width horizontal edge (reaches both borders)
    img[:, : w // 5] = (0.2, 0.3, 0.6)                                    # full-height vertical edge
    for k in range(6):                                                    # half-planes at assorted slopes
        ang = rng.uniform(0, np.pi)
        cx, cy = rng.uniform(0.2, 0.8) * w, rng.uniform(0.2, 0.8) * h
        m = ((xx - cx) * np.cos(ang) + (yy - cy) * np.sin(ang)) > 0
        box = (np.abs(xx - cx) < w * 0.18) & (np.abs(yy - cy) < h * 0.22)
        img[m & box] = rng.uniform(0, 1, 3)
    for k in range(5):                                                    # discs and rings
        cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(3, max(3.5, min(w, h) * 0.15))
        d = np.hypot(xx - cx, yy - cy)
        img[d < r] = rng.uniform(0, 1, 3)
        img[np.abs(d - 1.6 * r) < 0.7] = rng.uniform(0.5, 1, 3)           # thin ring: one-pixel features
    cb = ((xx.astype(int) + yy.astype(int)) & 1).astype(bool)
    reg = (xx > w * 0.62) & (xx < w * 0.74) & (yy > h * 0.05) & (yy < h * 0.25)
    img[reg & cb] = 0.9                                                   # checkerboard: edges everywhere
    img[reg & ~cb] = 0.1
    for k, step in enumerate((0.03, 0.049, 0.051, 0.099, 0.101, 0.149, 0.151, 0.3)):   # luma steps around the thresholds
        x0 = int(w * 0.78) + 3 * k
        img[int(h * 0.55):int(h * 0.9), x0:x0 + 3] = 0.3 + (step if k % 2 else -step) * 0 + step * (k + 1) / (k + 1)
    img[int(h * 0.55):int(h * 0.9), int(w * 0.78) + 24:] = 0.3
    stair = (yy - h * 0.5) > 0.37 * (xx - w * 0.3)                        # a long shallow staircase
    band = (xx > w * 0.25) & (xx < w * 0.6) & (yy > h * 0.38) & (yy < h * 0.62)
    img[stair & band] = (0.9, 0.85, 0.2)
    nz = (xx < w * 0.15) & (yy < h * 0.2)
    img[nz] = rng.uniform(0, 1, (int(nz.sum()), 3))                       # noise patch
    out = np.empty((h, w, 4), np.uint8)
    out[..., :3] = np.clip(img * 255.0 + 0.5, 0, 255).astype(np.uint8)
    out[..., 3] = 255
    return out
