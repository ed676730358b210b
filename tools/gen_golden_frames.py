#!/usr/bin/env python3
"""Generates tests/golden/frame_*.npz: small float32 frames rendered by the ORACLE, together with
every input needed to reproduce them (scene blocks as an RTXB blob, rt_defines, the tiny textures).

The reference ships no golden images; these fixtures pin the oracle's own
output against silent drift and give the GPU tests a committed vector to hit. Regenerate only on
purpose (after a reviewed change of the oracle or of the texture rule) and say so in the commit.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from raytracing_opengl_amd import scenes, textures  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
CASES = [
    ("frame_default_160x90_d4", dict(kind="default", width=160, height=90, depth=4)),
    ("frame_default_moved_128x96_d5", dict(kind="default", width=128, height=96, depth=5, time=6.5, delta=0.9, yaw=48.0, pitch=-6.0, cam_pos=(2.5, 1.0, -3.5))),
    ("frame_quadric_128x72_d4", dict(kind="quadric", width=128, height=72, depth=4)),
    ("frame_torus_96x54_d6", dict(kind="torus", width=96, height=54, depth=6)),
]


def main():
    ts = textures.default_texture_set(scale=64)  # 64x32 planets, 128x7 ring, 8x8 crate, 32x32 sky faces
    for name, kw in CASES:
        sc = scenes.build_scene(**kw)
        w, h = kw["width"], kw["height"]
        data = {"defines": np.array(sc.defines, dtype=np.float64), "width": w, "height": h}
        for bname, blob in sc.blocks.items():
            data["block_" + bname] = np.frombuffer(blob, dtype=np.uint8)
        for uniform, unit, img in ts["textures"]:
            data["tex_%d_%s" % (unit, uniform)] = img
        for f, face in enumerate(ts["cubemap"]):
            data["sky_%d" % f] = face
        for lod in (0, 1):
            img, cnt = oracle.OracleScene(sc, w, h, ts["textures"], ts["cubemap"], texture_lod=lod).render()
            data["frame_lod%d" % lod] = img
            data["rays_lod%d" % lod] = np.array([cnt["rays_closest"], cnt["rays_shadow"]], dtype=np.int64)
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **data)
        print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
