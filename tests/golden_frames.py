"""Loader for tests/golden/frame_*.npz (made by tools/gen_golden_frames.py)."""
import glob
import os

import numpy as np

from raytracing_opengl_amd.scenes import BLOCK_NAMES, SceneBlocks

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(glob.glob(os.path.join(GOLDEN, "frame_*.npz")))


def load(path):
    z = np.load(path)
    d = z["defines"]
    defines = tuple(int(v) for v in d[:9]) + tuple(float(np.float32(v)) for v in d[9:15])
    blocks = {name: z["block_" + name].tobytes() if ("block_" + name) in z.files else b"" for name in BLOCK_NAMES}
    sc = SceneBlocks(defines=defines, blocks=blocks)
    tex = []
    for key in z.files:
        if key.startswith("tex_"):
            _t, unit, uniform = key.split("_", 2)
            tex.append((uniform, int(unit), z[key]))
    sky = [z["sky_%d" % f] for f in range(6)]
    return dict(scene=sc, width=int(z["width"]), height=int(z["height"]), textures=tex, cubemap=sky,
                frames={0: z["frame_lod0"], 1: z["frame_lod1"]}, rays={0: tuple(z["rays_lod0"]), 1: tuple(z["rays_lod1"])})
