#!/usr/bin/env python3
"""Run the REFERENCE's own fragment shader (read from /root/reference at run time) on Mesa llvmpipe for the cases of
tests/reference_frames.py and write tests/golden/ref_frame_<name>.npz. Build container only (needs /root/reference and
the Mesa software rasteriser); the fixtures it writes are data: scene blocks + the reference's output pixels.

llvmpipe is asked for its precise paths: GALLIVM_PERF=no_aos_sampling,no_quad_lod (float texture filtering instead of
8-bit fixed point, per-pixel instead of per-quad level of detail)."""
import os
import sys

os.environ.setdefault("GALLIVM_PERF", "no_aos_sampling,no_quad_lod")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import reference_frames as rf  # noqa: E402
from oracle import oracle  # noqa: E402
from oracle.ref_gl import ref_gl  # noqa: E402
from raytracing_opengl_amd.scenes import BLOCK_NAMES  # noqa: E402


def main():
    if not ref_gl.available():
        sys.exit("needs /root/reference and Mesa's swrast_dri.so (build container only)")
    print("GL:", ref_gl.renderer())
    only = sys.argv[1:]
    cases = dict(rf.CASES)
    for vname, (case, _variant) in rf.VARIANTS.items():
        cases[vname] = (rf.CASES[case][0], True, rf.SAME_MIPS[1:])
    for name, (build, textured, limits) in cases.items():
        if only and name not in only:
            continue
        sc = build()
        ts = rf.texture_set(name)
        same_mips = name.endswith("_same_mips")
        level0 = name.endswith("_level0")
        W, H = rf.size(name)
        cube_mips = rf.cube_mipmap(name)
        ref, inactive = ref_gl.render(sc, W, H, ts["textures"], ts["cubemap"], cube_mipmap=cube_mips, oracle_mips=same_mips, level0_only=level0)
        arrays = dict(width=W, height=H, frame=np.ascontiguousarray(ref[..., :3]), digest=rf.input_digest(sc, ts),
                      renderer=ref_gl.renderer() + " GALLIVM_PERF=" + os.environ["GALLIVM_PERF"], tex_scale=rf.TEX_SCALE,
                      defines=np.asarray(sc.defines, dtype=np.float64), inactive_blocks=",".join(sorted(inactive)))
        for n in BLOCK_NAMES:
            if sc.blocks.get(n):
                arrays["block_" + n] = np.frombuffer(sc.blocks[n], dtype=np.uint8)
        if name in rf.PRIMARY_HITS:
            # a second run of the reference's shader, instrumented at run time to output the first calcInter's (t, type, num): the ROOT
            # Durand-Kerner accepted (and every other primitive's distance), for tests that compare roots instead of colours
            hit, _ = ref_gl.render(sc, W, H, ts["textures"], ts["cubemap"], patch=ref_gl.instrument_primary_hit)
            arrays["primary_t"] = np.ascontiguousarray(hit[..., 0])
            arrays["primary_type"] = hit[..., 1].astype(np.int8)
            arrays["primary_num"] = hit[..., 2].astype(np.int16)
        if textured and not same_mips and not level0 and not cube_mips:
            # the plain run: glGenerateMipmap built the mip levels, and its filter is the implementation's. Keep what llvmpipe built --
            # as the difference from the oracle's integer-mean levels (0 or +-1 nearly everywhere: compresses to a few KB) -- so that the
            # frame can also be judged with the SAME mip texels on both sides (tests/reference_classify.py, gl_mips)
            O = oracle.OracleScene(sc, W, H, ts["textures"], ts["cubemap"])
            for uniform, _unit, img in ts["textures"]:
                ours, theirs = O.mip_levels(uniform), ref_gl.generated_mips(img)
                assert len(ours) == len(theirs), (uniform, len(ours), len(theirs))
                for L, (a, b) in enumerate(zip(ours, theirs), start=1):
                    assert a.shape == b.shape
                    arrays[f"glmip_{uniform}_{L}"] = (b.astype(np.int16) - a.astype(np.int16)).astype(np.int8)
        np.savez_compressed(rf.path(name), **arrays)
        img, _ = oracle.OracleScene(sc, W, H, ts["textures"], ts["cubemap"], texture_lod=2 if same_mips else (0 if level0 else 1), cube_mipmap=cube_mips).render(0, H, threads=os.cpu_count() or 1)
        f4, f2, mx = rf.compare(img, ref[..., :3])
        assert np.all(ref[..., 3] == 1.0)
        print(f"{name:34s} oracle vs reference: {100*f4:7.3f}% of pixels > 1e-4, {100*f2:7.3f}% > 1e-2, max {mx:.3g}   "
              f"(limits {100*limits[0]:.2f}% / {100*limits[1]:.2f}%)  {os.path.getsize(rf.path(name))//1024} KB")


if __name__ == "__main__":
    main()
