"""GPU box: the default scene at 4K on the default and on the many-primitive build of the trace kernel (RTX_OPT_HIGH_OCCUPANCY = -1 / 0 / 1),
kernel time + frame checksum; with RTX_HIP_LIB pointing at a variant whose RT_PENCIL_MIN_PRIMS (rt_scene_dev.h) and RT_LANE_DIVERGENT_MIN were
set to 1 it showed what ray pencils cost a scene with three long-table primitives (profiles/r03_experiments.txt)."""
import os, sys, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np
from raytracing_opengl_amd import scenes, textures, wrapper
ts = textures.default_texture_set()
sc = scenes.build_scene("default", 3840, 2160, 4)
for occ in (-1, 0, 1):
    gl = wrapper.make_renderer(sc, 3840, 2160, ts["textures"], ts["cubemap"])
    gl.set_option(wrapper.RTX_OPT_HIGH_OCCUPANCY, occ)
    for _ in range(3): gl.draw()
    gl.finish()
    best = 1e9
    for _ in range(3):
        for _ in range(20): gl.draw()
        gl.finish(); best = min(best, gl.sum_recent_draw_ms(20) / 20)
    st = gl.stats()
    f = gl.read_pixels()
    print(os.environ.get("RTX_HIP_LIB", "product")[-24:], "occ", occ, f"{best*1000:.1f} us", "variant", st["kernel_variant"], "tables", st["candidate_tables"], "pencils", st["pencils"], hashlib.sha1(np.ascontiguousarray(f).tobytes()).hexdigest()[:10], flush=True)
    gl.stop()
