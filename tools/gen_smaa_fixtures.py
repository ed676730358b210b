"""Produces tests/golden/smaa_ref_*.npz: the reference's OWN SMAA shaders (assets/shaders/SMAA.h assembled as SMAA_Builder.h does)
executed on Mesa llvmpipe (oracle/ref_gl/ref_smaa.py) for a set of input frames. Build container only (needs /root/reference).

Each file holds the input frame, the preset, and the three textures the reference's draw() leaves behind (edges, weights, screen).
The look-up tables are the ones of tests/smaa_tables.py (search: equal to the reference's; area: synthetic) so that the vectors can
be replayed anywhere without the reference's AreaTex.h; tests/test_smaa_oracle.py additionally runs the live comparison with the
reference's real tables where the checkout exists."""
import os
import sys

os.environ.setdefault("GALLIVM_PERF", "no_aos_sampling,no_quad_lod")   # float texture filtering, like the tracer's reference frames
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402

import smaa_cases  # noqa: E402
import smaa_tables  # noqa: E402
from oracle.ref_gl import ref_gl, ref_smaa  # noqa: E402

CASES = [("default", "ULTRA", lambda: smaa_cases.traced("default", 256, 144, 4)),
         ("default", "LOW", lambda: smaa_cases.traced("default", 256, 144, 4)),
         ("torus", "HIGH", lambda: smaa_cases.traced("torus", 224, 126, 6)),
         ("quadric", "MEDIUM", lambda: smaa_cases.traced("quadric", 224, 126, 4)),
         ("pattern", "ULTRA", lambda: smaa_cases.pattern(1, 320, 200)),
         ("pattern", "HIGH", lambda: smaa_cases.pattern(2, 203, 131)),
         ("pattern", "MEDIUM", lambda: smaa_cases.pattern(3, 160, 100)),
         ("pattern", "LOW", lambda: smaa_cases.pattern(4, 131, 77))]


def main():
    area, search = smaa_tables.area_table(), smaa_tables.search_table()
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, preset, make in CASES:
        color = make()
        r = ref_smaa.run(color, preset, area, search)
        path = os.path.join(out_dir, f"smaa_ref_{name}_{preset.lower()}.npz")
        np.savez_compressed(path, color=color, preset=preset, edges=r["edges"], blend=r["blend"], screen=r["screen"],
                            renderer=ref_gl.renderer() + " GALLIVM_PERF=" + os.environ["GALLIVM_PERF"])
        print(path, color.shape, "edge px", int((r["edges"] != 0).any(-1).sum()), os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
