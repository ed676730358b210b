#!/usr/bin/env python3
"""GPU box, diagnostic: one random scene with a mip-mapped sky box -- which pixels differ between the HIP kernel and the oracle, and does
the kernel's value equal the oracle's at some forced level? usage: debug_cube_seed.py generator seed width height"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import random_scenes  # noqa: E402
from oracle import oracle  # noqa: E402
from raytracing_opengl_amd import textures, wrapper  # noqa: E402

gen, seed, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ts = textures.default_texture_set(scale=16)
sc = getattr(random_scenes, gen)(seed, w, h)
O = oracle.OracleScene(sc, w, h, ts["textures"], ts["cubemap"], texture_lod=1, cube_mipmap=True)
tags = np.zeros((h, w), np.uint32)
ref, cnt = O.render(tags=tags)
gl = wrapper.make_renderer(sc, w, h, ts["textures"], ts["cubemap"], cube_mipmap=True)
gl.draw()
img = gl.read_pixels()
gl.set_option(wrapper.RTX_OPT_CULL, 0)
gl.draw()
img_nocull = gl.read_pixels()
gl.stop()
print("culls on == off:", np.array_equal(img.view(np.uint32), img_nocull.view(np.uint32)))
with np.errstate(invalid="ignore", over="ignore"):
    d = np.abs(img - ref) / np.maximum(1.0, np.abs(ref))
d = np.where(np.isfinite(d), d, 0.0).max(-1)
ys, xs = np.nonzero(d > 1e-4)
print(len(ys), "pixels over the bar; tags of them:", np.unique(tags[ys, xs]))
forced = [O.render(lod_force=float(l))[0] for l in range(0, 9)]
flat = oracle.OracleScene(sc, w, h, ts["textures"], ts["cubemap"], texture_lod=1, cube_mipmap=False).render()[0]
for y, x in list(zip(ys, xs))[:12]:
    m = [float(np.abs(f[y, x] - img[y, x]).max() / max(1.0, np.abs(ref[y, x]).max())) for f in forced]
    print(f"({x},{y}) gpu {img[y, x, :3]} oracle {ref[y, x, :3]} tag {tags[y, x]}; quad tags {tags[y & ~1:(y & ~1) + 2, x & ~1:(x & ~1) + 2].ravel()}; "
          f"|gpu - oracle(level L)| rel: {['%.1e' % v for v in m]}; |gpu - no-mips| {float(np.abs(flat[y, x] - img[y, x]).max()):.2e}")
