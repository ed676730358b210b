#!/usr/bin/env python3
"""GPU box: render fuzz scenes with every kernel variant / option set (counting variant, product variant with and without ray pencils,
light variant, literal scans) and report which frames differ from the literal scans, with the first differing pixels.
usage: tools/variant_diff.py <generator> <seed> [seed ...]     (generators: tests/random_scenes.py)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import random_scenes
from raytracing_opengl_amd import textures, wrapper
ts = textures.default_texture_set(scale=16)
sizes = [(160, 96), (161, 97), (323, 181), (97, 161), (200, 120)]
gen = sys.argv[1]
for seed in map(int, sys.argv[2:]):
    w, h = sizes[seed % len(sizes)]
    sc = getattr(random_scenes, gen)(seed, w, h)
    frames = {}
    for name, opts in (("count", {wrapper.RTX_OPT_COUNT_RAYS: 1}), ("pen", {}), ("nopen", {wrapper.RTX_OPT_RAY_PENCILS: 0}),
                       ("light", {wrapper.RTX_OPT_HIGH_OCCUPANCY: 0}), ("nocull", {wrapper.RTX_OPT_CULL: 0})):
        gl = wrapper.make_renderer(sc, w, h, ts["textures"], ts["cubemap"])
        for k, v in opts.items():
            gl.set_option(k, v)
        gl.draw()
        frames[name] = gl.read_pixels()
        if name == "pen": print("  pencils", gl.stats()["pencils"])
        gl.stop()
    print(gen, seed, (w, h), sc.defines[:9])
    base = frames["nocull"].view(np.uint32)
    for name, f in frames.items():
        d = (f.view(np.uint32) != base).any(-1)
        ys, xs = np.nonzero(d)
        print(f"  {name:7s} differs from nocull in {int(d.sum())} pixels", [(int(x), int(y)) for x, y in zip(xs[:6], ys[:6])])
        for x, y in list(zip(xs, ys))[:2]:
            print("     ", f[y, x], frames["nocull"][y, x])
