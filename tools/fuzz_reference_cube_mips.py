#!/usr/bin/env python3
"""Build container only: the cube-mip rule ("cube mips") against the REFERENCE's shader on random content. Random scenes
(tests/random_scenes.py, object textures switched off: the sky is the only mip-mapped fetch) are run through rt.frag on Mesa llvmpipe with the
sky box loaded as GLWrapper::load_cubemap(faces, genMipmap = true) loads it (GLWrapper.cpp:307-310), and the oracle's frame goes through the
pixel-by-pixel accounting of tests/reference_classify.py (oracle in llvmpipe's level formula, `texture` class held at 5e-3 with band-limited
sky faces). Prints the classes' totals and every scene that leaves a pixel unexplained.
usage: [FUZZ_NASTY=1] tools/fuzz_reference_cube_mips.py first_seed count"""
import os
import sys

os.environ.setdefault("GALLIVM_PERF", "no_aos_sampling,no_quad_lod")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import random_scenes  # noqa: E402
import reference_classify as rc  # noqa: E402
import reference_frames as rf  # noqa: E402
from oracle.ref_gl import ref_gl  # noqa: E402
from raytracing_opengl_amd import textures  # noqa: E402


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    w, h = 112, 64
    ts = textures.default_texture_set(scale=16, smooth_sky=True)
    total, bad = {}, []
    for seed in range(first, first + count):
        gen = random_scenes.nasty_scene if os.environ.get("FUZZ_NASTY") else random_scenes.random_scene
        sc = rf._strip_textures(gen(seed, w, h))
        frame, _ = ref_gl.render(sc, w, h, ts["textures"], ts["cubemap"], cube_mipmap=True)
        ref = dict(name=f"fuzz_cube_{seed}", scene=sc, width=w, height=h, textures=ts["textures"], cubemap=ts["cubemap"], cube_mipmap=True,
                   frame=np.ascontiguousarray(frame[..., :3]), gl_mips=None)
        r = rc.classify(ref, texture_lod=2, tex_tol=5e-3, tex_level_envelope=True)
        for k, v in r.items():
            if isinstance(v, int):
                total[k] = total.get(k, 0) + v
        if r["unexplained"]:
            bad.append((seed, r["unexplained"], r["where"][:3]))
        rc._PROBES.clear(); rc._SAMPLES.clear()
    print(f"{count} scenes from seed {first} ({'nasty' if os.environ.get('FUZZ_NASTY') else 'random'}), {w}x{h}: " + ", ".join(f"{k} {v}" for k, v in total.items()))
    for b in bad:
        print("  unexplained:", b)


if __name__ == "__main__":
    main()
