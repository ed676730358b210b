#!/usr/bin/env python3
"""Diagnostic (GPU box, -DSMAA_PHASE_TIMES build): where do the waves of smaa_weights_kernel spend their time? One ULTRA resolve of the
traced 4K frame; per wave s_memtime stamps at the convergent points of the weight computation (100 MHz counter: 10 ns ticks)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

from raytracing_opengl_amd import _capi, scenes, textures, wrapper  # noqa: E402


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "ULTRA"
    w, h = 3840, 2160
    lib = _capi.load()
    fn = lib.rtx_debug_smaa_phase_times
    fn.argtypes = [ctypes.c_void_p]
    ts = textures.default_texture_set(scale=1)
    sc = scenes.build_scene("default", w, h, 4)
    gl = wrapper.make_renderer(sc, w, h, ts["textures"], ts["cubemap"])
    gl.draw()
    traced = gl.read_pixels(wrapper.RTX_RGBA8)
    gl.enable_SMAA(preset)
    gl.write_pixels(traced)
    for _ in range(4):
        gl.smaa_resolve()
    gl.finish()
    buf = np.zeros((8192, 12), dtype=np.uint64)[:8192]
    assert fn(buf.ctypes.data) == 0
    t = buf.astype(np.float64)
    t0 = t[:, 0].min()
    work = t[:, 8] > 0
    print(f"{preset}: {int(work.sum())} of 4096 waves had a pixel; kernel span (first entry -> last exit) {(t[:, 7].max() - t0) * 0.01:.1f} us; ticks are 10 ns")
    order = ["entry", "list prefix (counter loads, scan, barrier)", "list entry", "own texel", "diagonal searches + weights", "x searches + north weights", "y searches + west weights", "store + exit"]
    idx = [0, 5, 6, 1, 2, 3, 4, 7]
    tw = t[work]
    tot = tw[:, 7] - tw[:, 0]
    slow = tot >= np.percentile(tot, 95)
    print(f"  wave lifetimes: median {np.median(tot) * 0.01:.1f} us, p95 {np.percentile(tot, 95) * 0.01:.1f}, max {tot.max() * 0.01:.1f}; entry of the working waves after the first: median {np.median(tw[:, 0] - t0) * 0.01:.1f} us, max {(tw[:, 0] - t0).max() * 0.01:.1f}")
    for name, sel in (("all working waves", np.ones(len(tw), bool)), ("slowest 5 %", slow)):
        print(f"  {name}: mean us per phase")
        for k in range(1, len(idx)):
            d = (tw[sel, idx[k]] - tw[sel, idx[k - 1]]) * 0.01
            print(f"    {order[k]:45s} {d.mean():6.2f}  (max {d.max():6.2f})")
    dsel = tw[:, 9] > tw[:, 1]          # waves in which some lane ran the diagonal searches ([9] is stamped inside that branch)
    if dsel.any():
        a = (tw[dsel, 9] - tw[dsel, 1]) * 0.01
        b = (tw[dsel, 2] - tw[dsel, 9]) * 0.01
        print(f"  diagonal phase split ({int(dsel.sum())} waves): the four searches mean {a.mean():.2f} (p95 {np.percentile(a, 95):.2f}, max {a.max():.2f}), crossing edges + area {b.mean():.2f} (p95 {np.percentile(b, 95):.2f}, max {b.max():.2f})")
    gl.stop()


if __name__ == "__main__":
    main()
