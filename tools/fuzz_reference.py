#!/usr/bin/env python3
"""Build container only: random scenes (tests/random_scenes.py, textures switched off) through the REFERENCE's shader on
llvmpipe (oracle/ref_gl) against the oracle. Prints, per scene class, how many pixels differ by more than 1e-4 / 1e-2.
usage: tools/fuzz_reference.py first_seed count"""
import os
import struct
import sys

os.environ.setdefault("GALLIVM_PERF", "no_aos_sampling,no_quad_lod")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import random_scenes  # noqa: E402
import reference_frames as rf  # noqa: E402
from oracle import oracle  # noqa: E402
from oracle.ref_gl import ref_gl  # noqa: E402
from raytracing_opengl_amd import textures  # noqa: E402


def refractive_box(sc) -> bool:
    b = sc.blocks.get("boxes_buf", b"")
    return any(struct.unpack_from("<f", b, i * 112 + 36)[0] > 0.0 for i in range(len(b) // 112))   # rt_material.refraction


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    w, h = 112, 64
    ts = textures.default_texture_set(scale=16)
    groups = {}
    for seed in range(first, first + count):
        gen = random_scenes.nasty_scene if os.environ.get("FUZZ_NASTY") else random_scenes.random_scene
        sc = rf._strip_textures(gen(seed, w, h))
        ref, _ = ref_gl.render(sc, w, h, ts["textures"], ts["cubemap"])
        img, cnt = oracle.OracleScene(sc, w, h, ts["textures"], ts["cubemap"], texture_lod=1).render()
        both_nan = np.isnan(img[..., :3]) & np.isnan(ref[..., :3])   # NaN on both sides = agreement
        f4, f2, mx = rf.compare(np.where(both_nan, 0.0, img[..., :3]), np.where(both_nan, 0.0, ref[..., :3]))
        key = ("refractive box" if refractive_box(sc) else ("torus" if sc.defines[4] else "other"))
        g = groups.setdefault(key, [])
        g.append((f4, f2, mx, seed))
    for key, g in groups.items():
        a = np.array([(x[0], x[1]) for x in g])
        worst = max(g, key=lambda x: x[0])
        print(f"{key:15s}: {len(g):4d} scenes; pixels > 1e-4: mean {100*a[:,0].mean():.3f}% median {100*np.median(a[:,0]):.3f}% worst {100*worst[0]:.2f}% (seed {worst[3]}); "
              f"> 1e-2: mean {100*a[:,1].mean():.3f}%; scenes with no pixel > 1e-4: {(a[:,0]==0).sum()}")


if __name__ == "__main__":
    main()
