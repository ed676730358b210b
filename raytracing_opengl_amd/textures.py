"""Seeded synthetic stand-ins for the reference's texture assets.

The reference's default scene binds five 2-D textures and one cubemap (reference
src/main.cpp:137-153); their formats and sizes are listed in SURVEY.md Appendix E.  The image files
themselves are third-party art that does not travel to the GPU box, so benches and parity tests
use deterministic procedural images of the SAME formats and sizes.  The same bytes are fed to the
oracle and to the HIP tracer.

``scale`` divides every dimension (scale=1 -> reference sizes; tests use 4 or 8 to stay fast).
"""
from __future__ import annotations

import numpy as np

# (name, sampler uniform, texture unit, width, height, channels) -- SURVEY.md Appendix E
REFERENCE_TEXTURES = (
    ("8k_jupiter.jpg", "texture_sphere_1", 1, 4096, 2048, 3),
    ("8k_saturn.jpg", "texture_sphere_2", 2, 4096, 2048, 3),
    ("2k_mars.jpg", "texture_sphere_3", 3, 2048, 1024, 3),
    ("8k_saturn_ring_alpha.png", "texture_ring", 4, 8192, 500, 4),
    ("container.png", "texture_box", 5, 512, 512, 4),
)
CUBEMAP_FACE = 2048  # sb_nebula: 6 x RGB8 2048^2


def _upsampled_noise(rng: np.random.Generator, h: int, w: int, cell: int) -> np.ndarray:
    """Smooth value noise: random grid of (h/cell x w/cell) nodes, bilinearly upsampled, in [0,1]."""
    gh, gw = max(2, h // cell + 2), max(2, w // cell + 2)
    g = rng.random((gh, gw), dtype=np.float32)
    ys = np.linspace(0, gh - 1.001, h, dtype=np.float32)
    xs = np.linspace(0, gw - 1.001, w, dtype=np.float32)
    y0 = ys.astype(np.int32)
    x0 = xs.astype(np.int32)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def _to_u8(img: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(np.clip(img * 255.0 + 0.5, 0, 255).astype(np.uint8))


def planet(seed: int, w: int, h: int, tint=(1.0, 0.85, 0.7), smooth: bool = False) -> np.ndarray:
    """RGB8 equirect 'gas giant': latitude bands + three octaves of smooth noise + fine grain. smooth: band-limited -- no per-texel grain,
    no finest octave, gentler bands (a texel differs from its neighbours by a fraction of a grey level, so that a sample barely depends on
    WHICH mip level an implementation picks: the fixtures that hold the oracle to the reference at 5e-3, tests/reference_frames.py)."""
    rng = np.random.default_rng(seed)
    lat = np.linspace(0, np.pi, h, dtype=np.float32)[:, None]
    bands = 0.5 + 0.25 * np.sin(lat * (5.0 if smooth else 14.0)) + (0.0 if smooth else 0.1 * np.sin(lat * 37.0 + 1.3))
    n = 0.5 * _upsampled_noise(rng, h, w, max(1, h // 8)) + 0.3 * _upsampled_noise(rng, h, w, max(1, h // (12 if smooth else 32))) \
        + (0.0 if smooth else 0.2 * _upsampled_noise(rng, h, w, max(1, h // 128)))
    grain = np.float32(0.0) if smooth else rng.random((h, w), dtype=np.float32) * 0.06 - 0.03
    base = np.clip(0.65 * bands + 0.45 * (n - 0.5) + grain, 0, 1)
    img = np.stack([base * tint[0], base * tint[1], np.clip(base * tint[2] + 0.1 * (n - 0.5), 0, 1)], axis=-1)
    return _to_u8(img)


def ring(seed: int, w: int, h: int, smooth: bool = False) -> np.ndarray:
    """RGBA8 ring strip: radial (u) bands; alpha has fully transparent gaps and fully opaque bands. smooth: band-limited (24 / 60 knots
    instead of 48 / 700, ramps into the gap and the opaque band instead of steps, no per-texel noise)."""
    rng = np.random.default_rng(seed)
    u = np.linspace(0, 1, w, dtype=np.float32)
    coarse = np.interp(u, np.linspace(0, 1, 24 if smooth else 48), rng.random(24 if smooth else 48)).astype(np.float32)
    fine = np.interp(u, np.linspace(0, 1, 60 if smooth else 700), rng.random(60 if smooth else 700)).astype(np.float32)
    dens = np.clip(1.6 * coarse + 0.5 * fine - 0.6, 0, 1)
    if smooth:
        dens = dens * np.clip(np.abs(u - 0.61) / 0.06 - 0.5, 0, 1)                      # the gap: alpha 0 in its middle, ramps at its edges
        dens = np.maximum(dens, np.clip(1.5 - np.abs(u - 0.25) / 0.05, 0, 1))           # the opaque band likewise
    else:
        dens[(u > 0.58) & (u < 0.64)] = 0.0  # a "Cassini division": exact alpha 0
        dens[(u > 0.2) & (u < 0.3)] = 1.0    # an opaque band: exact alpha 255
    rgb = np.stack([0.85 * (0.6 + 0.4 * fine), 0.78 * (0.6 + 0.4 * fine), 0.62 * (0.6 + 0.4 * coarse)], axis=-1)
    img = np.concatenate([rgb, dens[:, None]], axis=-1)[None, :, :].repeat(h, axis=0)
    if not smooth:
        img = img + (rng.random((h, w, 1), dtype=np.float32) * 0.02 - 0.01) * (img > 0) * (img < 1)
    return _to_u8(img)


def crate(seed: int, w: int, h: int, smooth: bool = False) -> np.ndarray:
    """RGBA8 'container': planks with dark seams and a frame, alpha 255. smooth: the planks and the frame as sine ramps instead of steps,
    no one-texel seams."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    plank = ((x / max(1, w // 8)).astype(np.int32) % 2).astype(np.float32)
    seam = (np.minimum(x % max(1, w // 8), max(1, w // 8) - (x % max(1, w // 8))) < max(1, w // 128)).astype(np.float32)
    frame = ((x < w // 16) | (x >= w - w // 16) | (y < h // 16) | (y >= h - h // 16)).astype(np.float32)
    if smooth:
        plank = 0.5 + 0.5 * np.sin(x * (np.pi / max(1, w // 8)))
        seam = np.zeros_like(x)
        frame = np.clip(1.0 - np.minimum(np.minimum(x, w - 1 - x), np.minimum(y, h - 1 - y)) / max(1.0, w / 8.0), 0, 1)
    wood = 0.55 + 0.1 * plank + 0.15 * (_upsampled_noise(rng, h, w, max(1, h // 64)) - 0.5) + 0.06 * np.sin(y * 0.4)
    wood = wood * (1 - 0.6 * seam) * (1 - 0.35 * frame)
    img = np.stack([wood, wood * 0.72, wood * 0.45, np.ones_like(wood)], axis=-1)
    return _to_u8(img)


def nebula_face(seed: int, n: int, smooth: bool = False) -> np.ndarray:
    """RGB8 sky face: dim coloured clouds plus sparse stars. smooth: band-limited (brighter clouds of the coarse noise only, no stars), for
    the fixtures that pin mip-mapped sky fetches (load_cubemap(faces, true)) without an implementation's level choice mattering much."""
    rng = np.random.default_rng(seed)
    c1 = _upsampled_noise(rng, n, n, max(1, n // 6))
    c2 = _upsampled_noise(rng, n, n, max(1, n // (10 if smooth else 24)))
    if smooth:
        cloud = np.clip(0.8 * c1 + 0.5 * c2 - 0.3, 0, 1)
        return _to_u8(np.stack([0.75 * cloud + 0.05, 0.5 * cloud * c2 + 0.1, 0.8 * cloud * c1 + 0.1], axis=-1))
    cloud = np.clip(0.9 * c1 + 0.4 * c2 - 0.7, 0, 1)
    img = np.stack([0.55 * cloud + 0.02, 0.25 * cloud * c2 + 0.02, 0.7 * cloud * c1 + 0.04], axis=-1)
    stars = rng.random((n, n), dtype=np.float32) > 0.9993
    img[stars] = rng.random((int(stars.sum()), 1), dtype=np.float32) * 0.6 + 0.4
    return _to_u8(img)


def default_texture_set(scale: int = 1, seed: int = 2024, smooth: bool = False, smooth_sky: bool = False) -> dict:
    """Returns {'textures': [(uniform, unit, array HxWxC uint8)], 'cubemap': [6 arrays NxNx3 uint8]}. smooth: the band-limited variants of
    the planets, the ring and the crate (see planet)."""
    s = max(1, int(scale))
    out = []
    for i, (_name, uniform, unit, w, h, c) in enumerate(REFERENCE_TEXTURES):
        w, h = max(4, w // s), max(4, h // s)
        if uniform == "texture_ring":
            img = ring(seed + i, w, h, smooth)
        elif uniform == "texture_box":
            img = crate(seed + i, w, h, smooth)
        else:
            tint = ((1.0, 0.85, 0.7), (0.95, 0.9, 0.65), (1.0, 0.55, 0.4))[i]
            img = planet(seed + i, w, h, tint, smooth)
        assert img.shape == (h, w, c) and img.dtype == np.uint8
        out.append((uniform, unit, img))
    n = max(4, CUBEMAP_FACE // s)
    faces = [nebula_face(seed + 100 + f, n, smooth_sky) for f in range(6)]
    return {"textures": out, "cubemap": faces}
