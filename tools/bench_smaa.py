"""Times the SMAA resolve (smaa_kernel.hip) on the GPU: the tracer's own 4K frame of the default scene (and a synthetic pattern),
per preset, by HIP events (rtx_stats.last_smaa_ms). Prints JSON lines: time, edge pixels, algorithmic GB/s (W*H*4 B read + W*H*4 B
written per resolve) and its fraction of the 8 TB/s HBM peak. Run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402

import smaa_cases  # noqa: E402
import smaa_tables  # noqa: E402
from raytracing_opengl_amd import scenes, textures, wrapper  # noqa: E402


def main():
    w, h = 3840, 2160
    reps = int(os.environ.get("REPS", "20"))
    area, search = smaa_tables.area_table(), smaa_tables.search_table()
    ts = textures.default_texture_set(scale=1)
    sc = scenes.build_scene("default", w, h, 4)
    gl = wrapper.make_renderer(sc, w, h, ts["textures"], ts["cubemap"])
    gl.draw()
    traced = gl.read_pixels(wrapper.RTX_RGBA8)
    gl.set_smaa_tables(area, search)
    frames = {"traced default scene": traced, "synthetic pattern": np.tile(smaa_cases.pattern(41, 960, 540), (4, 4, 1))}
    only = os.environ.get("ONLY")          # e.g. ONLY=traced:ULTRA for a PMC pass of one configuration
    for name, img in frames.items():
        for preset in ("LOW", "MEDIUM", "HIGH", "ULTRA"):
            if only and only != f"{name.split()[0]}:{preset}":
                continue
            gl.enable_SMAA(preset)
            gl.write_pixels(img)
            times = []
            for _ in range(reps):
                gl.smaa_resolve()
                st = gl.stats()
                times.append(st["last_smaa_ms"])
            ms = float(np.median(times[2:]))
            algo = w * h * 8
            print(json.dumps({"frame": name, "size": [w, h], "preset": preset, "smaa_ms": round(ms, 4), "min_ms": round(min(times), 4),
                              "edge_pixels": st["smaa_edge_pixels"], "edge_fraction": round(st["smaa_edge_pixels"] / (w * h), 4),
                              "algorithmic_bytes": algo, "GBps": round(algo / ms / 1e6, 1), "frac_of_8TBps": round(algo / ms / 1e6 / 8000.0, 4)}))
    gl.stop()


if __name__ == "__main__":
    main()
