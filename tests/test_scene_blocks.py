"""Scene-description half of the boundary: this repo's scene.h / SceneManager / SurfaceFactory /
GLM-compat headers must produce byte-identical uniform blocks to the reference's own host code.

tests/golden/*.rtxb were generated IN THE BUILD CONTAINER by tools/gen_golden_blocks.sh, which
compiles the scene recipes against the reference's src/scene.h, src/Surface.h,
src/SceneManager.cpp and vendored GLM. The sha256 values are the independent pins of
SURVEY.md Appendix C.1 (made from the reference's main.cpp).
"""
import hashlib
import os
import struct

import pytest

from raytracing_opengl_amd import scenes

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")

CASES = [
    ("default_t0_1920x1080_d4.rtxb", dict(kind="default", width=1920, height=1080, depth=4)),
    ("default_t12.5_640x480_d1.rtxb", dict(kind="default", width=640, height=480, depth=1, time=12.5, delta=0.75, yaw=25.0, pitch=10.0)),
    ("quadric_3840x2160_d4.rtxb", dict(kind="quadric", width=3840, height=2160, depth=4)),
    ("torus_3840x2160_d6.rtxb", dict(kind="torus", width=3840, height=2160, depth=6)),
]

SURVEY_SHA256 = {  # SURVEY.md Appendix C.1, default scene t=0, canvas 1920x1080, depth 4
    "spheres_buf": "7f3590195ed112250df438c6699c23ffff7a875208edd5c3e81460684da11011",
    "surfaces_buf": "2293d8da11f835c14ba09143e614515ee39e8d3effdfd7838307537c798e9d9a",
    "boxes_buf": "c0ac4a8de61bd7a534ca4e4e3c611c266bbdd13ed2018b733cd5fcf074123015",
    "toruses_buf": "2ded90803df0b108d96139ee86aad84b3c2b2e756428bbd8b19fe26f89ecea90",
    "rings_buf": "5ee4c557bb34dff9a98f8716e9d5f62bb0d15f2673a35964da460c858da47d71",
    "lights_point_buf": "0ed56f45599f599b263846db234677dd203b518f484f3be417549cf90a472312",
    "lights_direct_buf": "5cd9789a4a0cc3b29d2677cbf0f51cb6cf41979b65c40b89c2cb58ce1f0ebe47",
    "scene_buf": "7f6c5a3cdc2f8973d181e44765aa16a17213bd4e08b14f4a323d64e0ce1eeb90",
}


@pytest.mark.parametrize("fname,kwargs", CASES)
def test_blocks_match_reference_host_code(built, fname, kwargs):
    golden = scenes.parse_rtxb(open(os.path.join(GOLDEN, fname), "rb").read())
    mine = scenes.build_scene(**kwargs)
    assert mine.defines == golden.defines
    for name in scenes.BLOCK_NAMES:
        assert mine.blocks[name] == golden.blocks[name], f"{fname}: block {name} differs from the reference host code"


def test_default_scene_matches_survey_pins(built):
    sc = scenes.build_scene("default", 1920, 1080, 4)
    for name, digest in SURVEY_SHA256.items():
        assert hashlib.sha256(sc.blocks[name]).hexdigest() == digest, name
    assert sc.blocks["planes_buf"] == b""
    assert sc.defines[:9] == (6, 0, 2, 2, 1, 1, 1, 1, 4)


def test_block_sizes_are_std140(built):
    sc = scenes.build_scene("quadric", 640, 360, 4)
    rec = dict(scene_buf=64, spheres_buf=112, planes_buf=96, surfaces_buf=160, boxes_buf=112, toruses_buf=112, rings_buf=112,
               lights_point_buf=48, lights_direct_buf=32)
    counts = dict(zip(("spheres_buf", "planes_buf", "surfaces_buf", "boxes_buf", "toruses_buf", "rings_buf", "lights_point_buf",
                       "lights_direct_buf"), sc.defines[:8]))
    for name, n in counts.items():
        assert len(sc.blocks[name]) == n * rec[name]
    assert counts["surfaces_buf"] == 96 and counts["planes_buf"] == 1
    # ring radii are stored squared (reference SceneManager.cpp:195-196)
    d = scenes.build_scene("default", 640, 480, 1)
    r1, r2 = struct.unpack_from("<2f", d.blocks["rings_buf"], 96)
    assert abs(r1 - (4150 * 1.1166) ** 2) / r1 < 1e-6 and abs(r2 - (4150 * 2.35) ** 2) / r2 < 1e-6


def test_odd_canvas_is_bumped_to_even(built):
    sc = scenes.build_scene("default", 641, 479, 1)  # reference main.cpp:39-41
    assert sc.canvas == (642, 480)
