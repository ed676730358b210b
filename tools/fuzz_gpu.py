#!/usr/bin/env python3
"""GPU box: many random scenes (tests/random_scenes.py) through the C ABI against the oracle -- max pixel difference and
ray counts -- beyond the seeds the test suite runs. Each scene is drawn twice: with the counting kernel variant (ray counts) and with the
PRODUCT variant (no counters; the many-primitive variant with its group culls where it is selected); both frames must be within the bar.
usage: [FUZZ_GEN=random_scene|nasty_scene|scaled_quat_scene|crowd_scene|pencil_scene] [FUZZ_CUBE_MIPS=1] tools/fuzz_gpu.py first_seed count [width height]
FUZZ_CUBE_MIPS: the sky box is loaded with genMipmap = true (GLWrapper.cpp:307-310) on both sides."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import parity_bar  # noqa: E402
import random_scenes  # noqa: E402
from oracle import oracle  # noqa: E402
from raytracing_opengl_amd import textures, wrapper  # noqa: E402


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    fixed = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else None
    sizes = [(160, 96), (161, 97), (323, 181), (97, 161), (200, 120)]   # odd sizes: helper invocations, axis-parallel centre rays
    ts = textures.default_texture_set(scale=16)
    worst, bad = 0.0, 0
    tally = dict(needed_relative=0, above_one=0, values=0)
    cube_mips = bool(os.environ.get("FUZZ_CUBE_MIPS"))
    ctx = {}   # one context per frame size, re-specialised per scene (creating a context costs ~0.2 s -- 40 GPU-minutes per 10 000 scenes)
    for seed in range(first, first + count):
        w, h = fixed or sizes[seed % len(sizes)]
        gen = os.environ.get("FUZZ_GEN", "nasty_scene" if os.environ.get("FUZZ_NASTY") else "random_scene")
        sc = getattr(random_scenes, gen)(seed, w, h)
        ref, cnt = oracle.OracleScene(sc, w, h, ts["textures"], ts["cubemap"], texture_lod=1, cube_mipmap=cube_mips).render()
        gl = ctx.get((w, h))
        if gl is None:
            gl = ctx[(w, h)] = wrapper.make_renderer(sc, w, h, ts["textures"], ts["cubemap"], cube_mipmap=cube_mips)
        else:
            gl.init_shaders(sc.defines)
            gl.uploader = wrapper.SceneUploader(sc, gl)
            gl.uploader.init()
        gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
        gl.draw()
        img = gl.read_pixels()
        st = gl.stats()
        gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 0)
        gl.draw()
        img_product = gl.read_pixels()
        nan_bad, mx = 0, 0.0
        for im in (img, img_product):
            # the bar (tests/parity_bar.py): 1e-4 absolute where |oracle| <= 1; a scene whose pixels reach 1e13 (non-unit quaternions scale
            # normals, pow() of values > 1) is judged relative to the pixel above 1 -- and how many values needed that is counted
            v = parity_bar.judge(im, ref)
            nan_bad += v["special_mismatch"]
            mx = max(mx, v["worst"])
            for k in tally:
                tally[k] += v[k]
        # the product variant (group culls, ray pencils) against the counting variant (first-level culls only): bit for bit
        variants_differ = int((img.view(np.uint32) != img_product.view(np.uint32)).any(-1).sum())
        nan_bad += variants_differ
        rays_ok = st["rays_closest"] == cnt["rays_closest"] and st["rays_shadow"] == cnt["rays_shadow"]
        worst = max(worst, mx)
        if nan_bad or mx > 1e-4 or not rays_ok:
            bad += 1
            print(f"seed {seed}: max (relative above 1) {mx:.3e} nan-mismatch {nan_bad} (pixels that differ between kernel variants: {variants_differ}) rays gpu {st['rays_closest']}+{st['rays_shadow']} oracle {cnt['rays_closest']}+{cnt['rays_shadow']}", flush=True)
        if (seed - first + 1) % 1000 == 0:
            print(f"... {seed - first + 1} scenes, {bad} outside the bar so far, worst {worst:.3e}", flush=True)
    print(f"{os.environ.get('FUZZ_GEN', 'random_scene')}: {count} scenes from seed {first} ({'%dx%d' % fixed if fixed else 'mixed sizes'}): {bad} outside the bar, worst max-abs difference {worst:.3e}")
    print(f"   the bar: 1e-4 absolute up to |oracle| = 1, relative above: {tally['values']} channel values judged (both kernel variants), {tally['above_one']} above 1, of which "
          f"{tally['needed_relative']} differ by more than 1e-4 absolute")


if __name__ == "__main__":
    main()
