#!/usr/bin/env python3
"""A/B timing on the GPU box: every library under raytracing_opengl_amd/variants/ (plus the product
library) renders the bench scenes at 4K; prints the mean kernel time and a frame checksum (all variants
must agree bit for bit). Each library runs in its own process (one HIP module per process)."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(scene_names, steps):
    import numpy as np
    from raytracing_opengl_amd import scenes, textures, wrapper
    ts = textures.default_texture_set()
    out = []
    for name in scene_names:
        name, _, depth = name.partition(":")          # "torus:6" = the torus scene at reflection depth 6 (default 4)
        W, H = (int(v) for v in os.environ.get("AB_SIZE", "3840x2160").split("x"))   # AB_SIZE=1920x1080: configs[1]
        sc = scenes.build_scene(name, W, H, int(depth or 4))
        gl = wrapper.make_renderer(sc, W, H, ts["textures"], ts["cubemap"])
        for _ in range(3):
            gl.draw()
        gl.finish()
        best = 1e9
        for _ in range(3):
            for _ in range(steps):
                gl.draw()
            gl.finish()
            best = min(best, gl.sum_recent_draw_ms(steps) / steps)
        frame = gl.read_pixels(wrapper.RTX_RGBA32F)
        out.append(f"{name} {best*1000:8.1f} us {hashlib.sha1(np.ascontiguousarray(frame).tobytes()).hexdigest()[:10]}")
        gl.stop()
    print(" | ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[3:], int(sys.argv[2]))
        sys.exit(0)
    scene_names = sys.argv[1:] or ["default", "quadric", "torus"]
    vdir = os.path.join(ROOT, "raytracing_opengl_amd", "variants")
    libs = [("product", os.path.join(ROOT, "raytracing_opengl_amd", "librtx_hip.so"))]
    libs += [(f[len("librtx_hip_"):-3], os.path.join(vdir, f)) for f in (sorted(os.listdir(vdir)) if os.path.isdir(vdir) else []) if f.endswith(".so") and "_dk" not in f and "_prof" not in f]
    for rep in range(2):
        for tag, path in libs:
            env = dict(os.environ, RTX_HIP_LIB=path)
            r = subprocess.run([sys.executable, __file__, "--child", os.environ.get("AB_STEPS", "20")] + scene_names, env=env, capture_output=True, text=True)
            print(f"{tag:16s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
