#!/usr/bin/env python3
"""Cost attribution by scene ablation (runs on the GPU box): time the 4K default-scene kernel with
one primitive group / light / feature removed at a time. No recompiles: only the uniform blocks and
rt_defines change."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import copy  # noqa: E402

from raytracing_opengl_amd import scenes, textures, wrapper  # noqa: E402

W, H, DEPTH = 3840, 2160, 4
IDX = dict(spheres_buf=0, planes_buf=1, surfaces_buf=2, boxes_buf=3, toruses_buf=4, rings_buf=5, lights_point_buf=6, lights_direct_buf=7)
REC = dict(spheres_buf=112, planes_buf=96, surfaces_buf=160, boxes_buf=112, toruses_buf=112, rings_buf=112, lights_point_buf=48, lights_direct_buf=32)


def variant(sc, drop=None, keep_first=None, depth=None):
    v = copy.deepcopy(sc)
    d = list(v.defines)
    for name in (drop or []):
        d[IDX[name]] = 0
        v.blocks[name] = b""
    for name, n in (keep_first or {}).items():
        d[IDX[name]] = n
        v.blocks[name] = v.blocks[name][: n * REC[name]]
    if depth is not None:
        d[8] = depth
    v.defines = tuple(d)
    return v


def time_scene(sc, ts, steps=int(os.environ.get("ABLATE_STEPS", "20"))):
    gl = wrapper.make_renderer(sc, W, H, ts["textures"], ts["cubemap"])
    gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    gl.draw()
    st = gl.stats()
    gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 0)
    for _ in range(3):
        gl.draw()
    gl.stats()
    for _ in range(steps):
        gl.draw()
    ms = gl.sum_recent_draw_ms(steps) / steps
    gl.stop()
    return ms, st["rays_closest"], st["rays_shadow"], st["rays_shadow_cast"]


def main():
    base = scenes.build_scene("default", W, H, DEPTH)
    ts = textures.default_texture_set(scale=int(os.environ.get("TEXSCALE", "1")))
    cases = [
        ("full default scene", base),
        ("depth 1", variant(base, depth=1)),
        ("no lights (no shadow rays)", variant(base, drop=["lights_point_buf", "lights_direct_buf"])),
        ("no point light", variant(base, drop=["lights_point_buf"])),
        ("no torus", variant(base, drop=["toruses_buf"])),
        ("no ring", variant(base, drop=["rings_buf"])),
        ("no surfaces", variant(base, drop=["surfaces_buf"])),
        ("no boxes", variant(base, drop=["boxes_buf"])),
        ("3 spheres (no planets)", variant(base, keep_first={"spheres_buf": 3})),
        ("no spheres", variant(base, drop=["spheres_buf"])),
        ("empty scene (sky only)", variant(base, drop=list(IDX))),
        ("empty scene, no sky texture", None),
    ]
    NL = ["lights_point_buf", "lights_direct_buf"]
    if os.environ.get("ABLATE_SERIES") == "closest":
        cases = [("closest-only d1: all", variant(base, drop=NL, depth=1))]
        for grp in ("spheres_buf", "surfaces_buf", "boxes_buf", "toruses_buf", "rings_buf"):
            cases.append((f"closest-only d1: no {grp}", variant(base, drop=NL + [grp], depth=1)))
        cases.append(("closest-only d1: spheres only", variant(base, drop=NL + ["surfaces_buf", "boxes_buf", "toruses_buf", "rings_buf"], depth=1)))
        cases.append(("closest-only d1: 3 spheres only", variant(base, drop=NL + ["surfaces_buf", "boxes_buf", "toruses_buf", "rings_buf"], keep_first={"spheres_buf": 3}, depth=1)))
        cases.append(("closest-only d1: boxes only", variant(base, drop=NL + ["surfaces_buf", "spheres_buf", "toruses_buf", "rings_buf"], depth=1)))
        cases.append(("closest-only d1: floor box only", variant(base, drop=NL + ["surfaces_buf", "spheres_buf", "toruses_buf", "rings_buf"], keep_first={"boxes_buf": 1}, depth=1)))
        cases.append(("closest-only d1: surfaces only", variant(base, drop=NL + ["boxes_buf", "spheres_buf", "toruses_buf", "rings_buf"], depth=1)))
        cases.append(("closest-only d1: torus only", variant(base, drop=NL + ["boxes_buf", "spheres_buf", "surfaces_buf", "rings_buf"], depth=1)))
        cases.append(("closest-only d1: ring only", variant(base, drop=NL + ["boxes_buf", "spheres_buf", "surfaces_buf", "toruses_buf"], depth=1)))
        cases.append(("closest-only d1: nothing", variant(base, drop=NL + ["boxes_buf", "spheres_buf", "surfaces_buf", "toruses_buf", "rings_buf"], depth=1)))
    if os.environ.get("ABLATE_SERIES") == "fixed":
        cases = [("empty scene, 0 iterations", variant(base, drop=list(IDX), depth=0)),
                 ("empty scene (sky only)", variant(base, drop=list(IDX))),
                 ("empty scene, no sky texture", None),
                 ("floor box only, no lights, d1", variant(base, drop=NL + ["surfaces_buf", "spheres_buf", "toruses_buf", "rings_buf"], keep_first={"boxes_buf": 1}, depth=1)),
                 ("floor box only, no lights, d4", variant(base, drop=NL + ["surfaces_buf", "spheres_buf", "toruses_buf", "rings_buf"], keep_first={"boxes_buf": 1})),
                 ("floor box only, 2 lights, d1", variant(base, drop=["surfaces_buf", "spheres_buf", "toruses_buf", "rings_buf"], keep_first={"boxes_buf": 1}, depth=1))]
    only = os.environ.get("ABLATE_ONLY")
    for name, sc in cases:
        if only and only not in name:
            continue
        if sc is None:
            sc2 = variant(base, drop=list(IDX))
            ms, rc, rs, rsc = time_scene(sc2, {"textures": [], "cubemap": None})
        else:
            ms, rc, rs, rsc = time_scene(sc, ts)
        print(f"{name:34s} {ms*1000:8.1f} us   rays closest {rc:9d} shadow {rs:9d} (cast {rsc})", flush=True)


if __name__ == "__main__":
    main()
