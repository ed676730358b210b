#!/usr/bin/env python3
"""How does the oracle's frame time scale with OpenMP threads on this host? (21x on 256 hardware threads.)
    python tools/time_oracle_threads.py [W H]      prints s per frame and Mray/s for 1 (a 1/16 sample), 8, 32, 64, 128, 256 threads"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from raytracing_opengl_amd import scenes, textures  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
sc = scenes.build_scene("default", W, H, 4)
ts = textures.default_texture_set()
for lod in (1, 0):
    o = oracle.OracleScene(sc, W, H, ts["textures"], ts["cubemap"], texture_lod=lod)
    o.render(0, 16, threads=8)      # mip chains, thread pool
    for n in [t for t in (1, 8, 16, 32, 64, 128, 256) if t <= (os.cpu_count() or 1)]:
        rows = (0, H) if n > 1 else (H // 2 - 32, H // 2 + 32)
        t0 = time.perf_counter()
        _img, cnt = o.render(rows[0], rows[1], threads=n)
        dt = time.perf_counter() - t0
        print(f"texture_lod {lod} threads {n:3d}: rows {rows[1] - rows[0]:4d}  {dt:7.3f} s  {cnt['rays'] / dt / 1e6:8.3f} Mray/s  ({cnt['rays'] / dt / 1e6 / n:.4f} per thread)", flush=True)
