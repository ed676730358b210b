#!/usr/bin/env python3
"""Diagnostic (GPU box, -DSMAA_PHASE_TIMES build): the role-split weight kernel (smaa_weights_roles_kernel) wave by wave. One resolve of the
traced 4K frame; s_memrealtime stamps (100 MHz: 10 ns ticks) at entry, list prefix, list entry + own texel, the wave's part, the barrier, exit."""
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
    fn = lib.rtx_debug_smaa_role_times
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
    buf = np.zeros((8192, 8), dtype=np.uint64)
    assert fn(buf.ctypes.data) == 0
    t = buf.astype(np.float64)
    live = t[:, 0] > 0
    t0 = t[live, 0].min()
    print(f"{preset}: kernel span (first entry -> last exit) {(t[live, 5].max() - t0) * 0.01:.1f} us; {int((t[:, 6] > 0).sum())} waves had a pixel; "
          f"entry of all waves after the first: median {np.median(t[live, 0] - t0) * 0.01:.2f}, max {(t[live, 0] - t0).max() * 0.01:.2f} us")
    idle = live & (t[:, 6] == 0)
    if idle.any():
        print(f"  waves without a pixel ({int(idle.sum())}): lifetime median {np.median(t[idle, 5] - t[idle, 0]) * 0.01:.2f}, max {(t[idle, 5] - t[idle, 0]).max() * 0.01:.2f} us")
    for role, name in enumerate(("diag 1", "north", "west", "diag 2")):
        sel = t[:, 6] == role + 1
        if not sel.any():
            continue
        a = t[sel]
        ran = a[:, 7] > 0
        d = lambda i, j, m=None: ((a[:, j] - a[:, i]) if m is None else (a[m, j] - a[m, i])) * 0.01   # noqa: E731
        part = d(2, 3, ran) if ran.any() else np.zeros(1)
        print(f"  {name:8s} {int(sel.sum()):5d} waves ({int(ran.sum())} ran the part): prefix {d(0, 1).mean():.2f} (max {d(0, 1).max():.2f}), entry + own texel {d(1, 2).mean():.2f} (max {d(1, 2).max():.2f}), "
              f"part mean {part.mean():.2f} p50 {np.median(part):.2f} p95 {np.percentile(part, 95):.2f} max {part.max():.2f}, wait at the barrier mean {d(3, 4).mean():.2f} max {d(3, 4).max():.2f}, "
              f"store + exit {d(4, 5).mean():.2f}; lifetime median {np.median(d(0, 5)):.2f} max {d(0, 5).max():.2f}; exit after the kernel's start: max {(a[:, 5].max() - t0) * 0.01:.2f}")
    gl.stop()


if __name__ == "__main__":
    main()
