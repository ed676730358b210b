"""Logic check without a GPU: the product's device header (rt_device.h) + scene packer
(rt_pack.h), compiled for the host by tests/host_harness, against the independently written
oracle. On the host both sides use the same libm, so agreement must be BIT-EXACT -- with the
conservative culls on and off (DESIGN.md "Culls"), and the ray counts must be identical.
The host build has no 2x2 quads, so these checks run the level-0 texture mode (texture_lod = 0);
the quad-derivative LOD mode is covered on the GPU (tests/test_gpu_parity.py)."""
import numpy as np
import pytest

import harness
from oracle import oracle
from raytracing_opengl_amd import scenes

CASES = [("default", 320, 240, 1), ("default", 256, 144, 4), ("quadric", 192, 108, 4), ("torus", 128, 72, 6)]


def test_unorm8_is_exact(built):
    assert harness.lib().harness_unorm8_mismatches() == 0


@pytest.mark.parametrize("kind,w,h,depth", CASES)
@pytest.mark.parametrize("cull", [False, True])
def test_device_logic_matches_oracle_bit_exactly(built, small_textures, kind, w, h, depth, cull):
    sc = scenes.build_scene(kind, w, h, depth)
    ref, cnt = oracle.OracleScene(sc, w, h, small_textures["textures"], small_textures["cubemap"], texture_lod=0).render()
    img, hc = harness.render(sc, w, h, small_textures["textures"], small_textures["cubemap"], cull=cull)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert hc["closest"] == cnt["rays_closest"] and hc["shadow_ref"] == cnt["rays_shadow"]
    assert hc["shadow_cast"] <= hc["shadow_ref"]
    if cull and cnt["dk_solves"]:
        assert hc["torus_solves"] < cnt["dk_solves"]


def test_animated_and_rotated_camera(built, small_textures):
    sc = scenes.build_scene("default", 200, 120, 5, time=7.0, delta=1.3, yaw=-35.0, pitch=12.0, cam_pos=(-6.0, 2.0, -3.0))
    ref, _ = oracle.OracleScene(sc, 200, 120, small_textures["textures"], small_textures["cubemap"], texture_lod=0).render()
    img, _ = harness.render(sc, 200, 120, small_textures["textures"], small_textures["cubemap"], cull=True)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


def test_unbound_samplers_and_missing_faces(built, small_textures):
    sc = scenes.build_scene("default", 160, 90, 2)
    faces = list(small_textures["cubemap"])
    faces[2] = None  # a face that failed to load stays black (reference GLWrapper.cpp:301-305)
    tex = [t for t in small_textures["textures"] if t[0] != "texture_box"]
    ref, _ = oracle.OracleScene(sc, 160, 90, tex, faces, texture_lod=0).render()
    img, _ = harness.render(sc, 160, 90, tex, faces, cull=True)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


def test_percent_f_round_trip_is_applied(built, small_textures):
    """T9 on the product side (GLWrapper.cpp:246-247,279-282: the two colour constants go through std::to_string = "%f"):
    with colours that the 6-decimal text CHANGES (0.1234567 -> 0.123457, 1e-7 -> 0, 0.0999999 -> 0.100000) the packer's frame
    must equal the oracle's bit for bit, and must equal the frame of the pre-rounded colours -- i.e. the digits beyond the
    sixth decimal cannot reach a pixel."""
    import ctypes
    import dataclasses
    w, h = 160, 90
    sc = scenes.build_scene("default", w, h, 3)
    raw = (0.1234567, 0.05000004, 1e-7, 0.3333333, 0.0999999, 0.2500001)
    rt = oracle.lib().orc_kat_text_round_trip
    pre = tuple(float(rt(ctypes.c_float(v))) for v in raw)
    assert [np.float32(a) != np.float32(b) for a, b in zip(raw, pre)].count(True) >= 5
    frames = []
    for cols in (raw, pre):
        s2 = dataclasses.replace(sc, defines=tuple(sc.defines[:9]) + cols)
        ref, _ = oracle.OracleScene(s2, w, h, small_textures["textures"], small_textures["cubemap"], texture_lod=0).render()
        img, _ = harness.render(s2, w, h, small_textures["textures"], small_textures["cubemap"], cull=True)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
        frames.append(img)
    assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))
    plain, _ = harness.render(sc, w, h, small_textures["textures"], small_textures["cubemap"], cull=True)
    assert not np.array_equal(plain.view(np.uint32), frames[0].view(np.uint32))
