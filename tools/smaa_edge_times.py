#!/usr/bin/env python3
"""Diagnostic (GPU box, -DSMAA_PHASE_TIMES build): when do the waves of the dense SMAA kernel (smaa_edges_kernel) start, how long do they
live, and in which phase? One ULTRA resolve of the traced 4K frame; s_memrealtime stamps (10 ns ticks) per wave."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

from raytracing_opengl_amd import _capi, scenes, textures, wrapper  # noqa: E402


def main():
    w, h = 3840, 2160
    lib = _capi.load()
    fn = lib.rtx_debug_smaa_edge_times
    fn.argtypes = [ctypes.c_void_p]
    ts = textures.default_texture_set(scale=1)
    sc = scenes.build_scene("default", w, h, 4)
    gl = wrapper.make_renderer(sc, w, h, ts["textures"], ts["cubemap"])
    gl.draw()
    traced = gl.read_pixels(wrapper.RTX_RGBA8)
    gl.enable_SMAA("ULTRA")
    gl.write_pixels(traced)
    for _ in range(4):
        gl.smaa_resolve()
    gl.finish()
    buf = np.zeros((8192, 6), dtype=np.uint64)
    assert fn(buf.ctypes.data) == 0
    t = buf.astype(np.float64)
    live = t[:, 0] > 0
    t = t[live]
    work = t[:, 5] > t[:, 1]
    t0 = t[:, 0].min()
    print(f"{len(t)} waves stamped, {int(work.sum())} with a strip; kernel span {(t[:, 5].max() - t0) * 0.01:.1f} us")
    tw = t[work]
    ent = (tw[:, 0] - t0) * 0.01
    life = (tw[:, 5] - tw[:, 0]) * 0.01
    print(f"  entry after the first wave: median {np.median(ent):.1f} us, p90 {np.percentile(ent, 90):.1f}, max {ent.max():.1f}")
    print(f"  lifetime: median {np.median(life):.1f} us, p95 {np.percentile(life, 95):.1f}, max {life.max():.1f};  exit: median {np.median((tw[:, 5] - t0) * 0.01):.1f}, max {((tw[:, 5] - t0) * 0.01).max():.1f}")
    names = ["luma tables + barrier", "first four rows loaded", "eight rows of edges + copy", "bit planes", "edge texels + append"]
    for k in range(5):
        d = (tw[:, k + 1] - tw[:, k]) * 0.01
        print(f"    {names[k]:32s} mean {d.mean():6.2f}  p95 {np.percentile(d, 95):6.2f}  max {d.max():6.2f}")
    late = tw[ent > np.percentile(ent, 90)]
    print(f"  the last tenth to start lives {np.median((late[:, 5] - late[:, 0]) * 0.01):.1f} us (median)")
    gl.stop()


if __name__ == "__main__":
    main()
