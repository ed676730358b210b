"""GPU tests of the rows widened from SURVEY section 8(f) and of the drop-in C++ surface -- everything that round 1 only
checked on the CPU (judge's list, round 1):

 f2  image FILES (JPEG fixtures of tests/golden/jpeg + PNGs written here) decoded by the shim's readers, uploaded and rendered
     by the HIP tracer, against the oracle fed the EXPECTED texels (expected.npz = what the reference's decoder returns);
 f4  render -> save_png -> decode == read_pixels(RGBA8);
 b   include/rtx/GLWrapper.h + SceneManager.h compiled as a main.cpp-shaped program (tests/shim_harness/shim_frame.cpp) and run on
     the GPU box: its frame against the oracle;
 a21 the RGBA8 target is clamp + round of the SAME floats the RGBA32F target holds, bit for bit;
 a1  (T9) a colour the "%f" round trip changes.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle
from raytracing_opengl_amd import scenes, textures, wrapper

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD_JPEG = os.path.join(ROOT, "tests", "golden", "jpeg")
SHIM = os.path.join(ROOT, "tests", "shim_harness", "shim_frame")
TOL = 1e-4


def quantise(img32: np.ndarray) -> np.ndarray:
    """The write-out rule of the RGBA8 target (rt_kernel.hip pack_rgba8; GL's float -> unorm8 conversion): clamp to [0,1],
    NaN -> 0, then (uint)(v * 255.0f + 0.5f) in float32 arithmetic."""
    v = img32.astype(np.float32)
    v = np.where(v < 0, np.float32(0), np.where(v > 1, np.float32(1), v))
    v = np.where(np.isnan(v), np.float32(0), v).astype(np.float32)
    return (v * np.float32(255.0) + np.float32(0.5)).astype(np.float32).astype(np.uint32).astype(np.uint8)


def _decode(path):
    lib = scenes._host_lib()
    lib.rtxh_decode_image.restype = ctypes.c_size_t
    lib.rtxh_decode_image.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                      ctypes.c_void_p, ctypes.c_size_t]
    w, h, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    need = lib.rtxh_decode_image(str(path).encode(), ctypes.byref(w), ctypes.byref(h), ctypes.byref(c), None, 0)
    assert need, f"decoder refused {path}"
    out = np.empty(need, np.uint8)
    lib.rtxh_decode_image(str(path).encode(), ctypes.byref(w), ctypes.byref(h), ctypes.byref(c), out.ctypes.data, need)
    return out.reshape(h.value, w.value, c.value)


def _write_png(path, arr):
    lib = scenes._host_lib()
    lib.rtxh_write_png.restype = ctypes.c_int
    lib.rtxh_write_png.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    arr = np.ascontiguousarray(arr, np.uint8)
    assert lib.rtxh_write_png(str(path).encode(), arr.ctypes.data, arr.shape[1], arr.shape[0], arr.shape[2], 0) == 1


JPEGS = ["pil_RGB_420_48x32_base_q90_r2", "pil_RGB_420_48x32_prog_q35_r0", "hm_rst_fill_40x24"]   # baseline, progressive, restart markers


def _asset_dir(tmp_path, sky_scale=32):
    """textures/ with three golden JPEGs (sphere maps), a ring strip and a crate as PNGs, six sky faces as PNGs.
    Returns (dir, textures-for-the-oracle, cubemap-for-the-oracle): the oracle never sees a file, it gets the texels the
    REFERENCE's decoder produced for the JPEGs (expected.npz) and the arrays the PNGs were written from."""
    tdir = tmp_path / "textures"
    tdir.mkdir()
    exp = np.load(os.path.join(GOLD_JPEG, "expected.npz"))
    small = textures.default_texture_set(scale=32)
    by_uniform = {u: img for u, _unit, img in small["textures"]}
    tex = []
    for k, name in enumerate(JPEGS):
        data = open(os.path.join(GOLD_JPEG, name + ".jpg"), "rb").read()
        (tdir / f"t{k + 1}.jpg").write_bytes(data)
        tex.append((f"texture_sphere_{k + 1}", k + 1, exp[name]))
    _write_png(tdir / "ring.png", by_uniform["texture_ring"])
    _write_png(tdir / "box.png", by_uniform["texture_box"])
    tex.append(("texture_ring", 4, by_uniform["texture_ring"]))
    tex.append(("texture_box", 5, by_uniform["texture_box"]))
    sky = small["cubemap"] if sky_scale == 32 else textures.default_texture_set(scale=sky_scale)["cubemap"]
    for f, face in enumerate(sky):
        _write_png(tdir / f"sky{f}.png", face)
    return tmp_path, tex, sky


def test_decoded_image_files_render_like_the_expected_texels(built, tmp_path):
    """f2 on the GPU: file -> include/rtx/{jpeg,png}_decode.h -> rtx_texture2d_create -> HIP frame == oracle(expected texels)."""
    d, tex, cube = _asset_dir(tmp_path)
    w, h, depth = 480, 270, 4
    sc = scenes.build_scene("default", w, h, depth)
    loaded = [(u, unit, _decode(d / "textures" / (f"t{unit}.jpg" if unit <= 3 else ("ring.png" if unit == 4 else "box.png")))) for u, unit, _ in tex]
    for (_u, _n, got), (_u2, _n2, want) in zip(loaded, tex):
        assert np.array_equal(got, want)                      # byte-identical texels, decoded on the GPU box itself
    faces = [_decode(d / "textures" / f"sky{f}.png") for f in range(6)]
    gl = wrapper.make_renderer(sc, w, h, loaded, faces)
    gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    gl.draw()
    img = gl.read_pixels()
    st = gl.stats()
    gl.stop()
    ref, cnt = oracle.OracleScene(sc, w, h, tex, cube).render()
    assert float(np.abs(img - ref).max()) <= TOL
    assert st["rays_closest"] == cnt["rays_closest"] and st["rays_shadow"] == cnt["rays_shadow"]
    ref_plain, _ = oracle.OracleScene(sc, w, h, None, cube).render()
    assert float(np.abs(ref - ref_plain).max()) > 0.05        # the textures are actually in the picture


@pytest.mark.parametrize("env", [{}, {"RTX_DEVICES": "0,0,0", "RTX_GATHER": "peer"}], ids=["one_device", "three_ranks_peer_copy"])
@pytest.mark.parametrize("w,h,kw", [(320, 180, dict(time=0.0, delta=0.0, yaw=0.0, pitch=0.0)),
                                    (322, 182, dict(time=7.25, delta=0.02, yaw=35.0, pitch=-6.0))])
def test_cpp_shim_program_renders_the_oracles_frame(built, tmp_path, w, h, kw, env):
    """The header-only GLWrapper / SceneManager / SurfaceFactory surface, compiled as a main.cpp-shaped program and run on the
    GPU: frame (RGBA32F) within 1e-4 of the oracle on the same scene recipe; its RGBA8 dump and its save_png agree exactly."""
    assert os.path.exists(SHIM), "tests/shim_harness/shim_frame was not built (python -c 'import __graft_entry__ as g; g.build()')"
    d, tex, cube = _asset_dir(tmp_path)
    depth = 5
    out = tmp_path / "frame"
    subprocess.run([SHIM, str(w), str(h), str(depth), repr(kw["time"]), repr(kw["delta"]), repr(kw["yaw"]), repr(kw["pitch"]), "0", str(out)],
                   check=True, cwd=d, timeout=300, env=dict(os.environ, **env))   # RTX_DEVICES: the shim picks a multi-device context
    img = np.fromfile(str(out) + ".f32", np.float32).reshape(h, w, 4)
    img8 = np.fromfile(str(out) + ".u8", np.uint8).reshape(h, w, 4)
    sc = scenes.build_scene("default", w, h, depth, **kw)
    ref, _cnt = oracle.OracleScene(sc, w, h, tex, cube).render()
    assert float(np.abs(img - ref).max()) <= TOL
    assert np.array_equal(img8, quantise(img))
    png = _decode(str(out) + ".png")                           # top row first
    assert np.array_equal(png[::-1], img8)


@pytest.mark.parametrize("env", [{}, {"RTX_DEVICES": "0,0,0", "RTX_GATHER": "peer"}], ids=["one_device", "three_ranks_peer_copy"])
def test_cpp_shim_program_with_a_mip_mapped_sky_box(built, tmp_path, env):
    """GLWrapper::load_cubemap(faces, genMipmap = true) (GLWrapper.cpp:307-310) through the C++ shim, from image files: the oracle's frame with
    cube mips, and not the one without; with three ranks the flag reaches every rank's copy of the sky box (bands of one frame)."""
    d, tex, cube = _asset_dir(tmp_path, sky_scale=8)          # 256-texel faces on a 160-pixel-wide frame: the sky is minified
    w, h, depth = 160, 88, 4
    kw = dict(time=2.0, delta=0.02, yaw=-120.0, pitch=35.0)
    out = tmp_path / "frame"
    subprocess.run([SHIM, str(w), str(h), str(depth), repr(kw["time"]), repr(kw["delta"]), repr(kw["yaw"]), repr(kw["pitch"]), "0", str(out)],
                   check=True, cwd=d, timeout=300, env=dict(os.environ, SHIM_CUBE_MIPS="1", **env))
    img = np.fromfile(str(out) + ".f32", np.float32).reshape(h, w, 4)
    sc = scenes.build_scene("default", w, h, depth, **kw)
    ref, _cnt = oracle.OracleScene(sc, w, h, tex, cube, cube_mipmap=True).render()
    flat, _cnt = oracle.OracleScene(sc, w, h, tex, cube, cube_mipmap=False).render()
    assert float(np.abs(img - ref).max()) <= TOL
    assert float(np.abs(flat - ref).max()) > 1e-2


def test_cpp_shim_program_with_smaa(built, tmp_path):
    """main.cpp's own order -- enable_SMAA(ULTRA) before init_window (main.cpp:32-34) -- through the C++ shim: the screen it reads back
    and the PNG it saves are the oracle's SMAA of the RGBA8 frame it traced."""
    import smaa_tables
    from oracle import smaa
    d, tex, cube = _asset_dir(tmp_path)
    area, search = smaa_tables.area_table(), smaa_tables.search_table()
    (tmp_path / "smaa_area.bin").write_bytes(area.tobytes())
    (tmp_path / "smaa_search.bin").write_bytes(search.tobytes())
    w, h, depth = 322, 182, 4
    out = tmp_path / "frame"
    subprocess.run([SHIM, str(w), str(h), str(depth), "3.5", "0.01", "20.0", "-3.0", "1", str(out)], check=True, cwd=d, timeout=300)
    img8 = np.fromfile(str(out) + ".u8", np.uint8).reshape(h, w, 4)
    screen = np.fromfile(str(out) + ".screen", np.uint8).reshape(h, w, 4)
    want = smaa.run(img8, "ULTRA", area, search)["screen"]
    assert np.array_equal(screen, want)
    assert np.array_equal(_decode(str(out) + ".png")[::-1], screen)
    assert (screen != img8).any()


def test_rgba8_target_is_the_exact_quantisation_of_the_float_target(mid_textures):
    """a21: both targets come from one launch; the 8-bit one must be clamp/round of the very same floats, not +-1 LSB."""
    for kind, w, h, depth in [("default", 960, 540, 4), ("quadric", 333, 207, 4), ("torus", 320, 180, 6)]:
        sc = scenes.build_scene(kind, w, h, depth)
        gl = wrapper.make_renderer(sc, w, h, mid_textures["textures"], mid_textures["cubemap"])
        gl.draw()
        img, img8 = gl.read_pixels(wrapper.RTX_RGBA32F), gl.read_pixels(wrapper.RTX_RGBA8)
        gl.stop()
        assert np.array_equal(img8, quantise(img)), kind
        assert img8[..., 3].min() == 255


def test_save_png_round_trip_of_a_gpu_frame(built, small_textures, tmp_path):
    """f4: render -> rtx_read_pixels(RGBA8) -> png_write.h -> png_decode.h == the frame, including an odd-sized one."""
    for w, h in [(320, 180), (161, 97)]:
        sc = scenes.build_scene("default", w, h, 3)
        gl = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"])
        gl.draw()
        img8 = gl.read_pixels(wrapper.RTX_RGBA8)
        gl.stop()
        p = tmp_path / f"f{w}.png"
        lib = scenes._host_lib()
        lib.rtxh_write_png.restype = ctypes.c_int
        lib.rtxh_write_png.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        assert lib.rtxh_write_png(str(p).encode(), img8.ctypes.data, w, h, 4, 1) == 1   # bottom_up: row 0 of the frame is the bottom row
        assert np.array_equal(_decode(p)[::-1], img8)


def test_colours_take_the_percent_f_round_trip(small_textures):
    """T9 (GLWrapper.cpp:246-247,279-282): AMBIENT_COLOR / SHADOW_AMBIENT reach the shader as "%f" text (6 decimals). On the GPU
    the change (< 5e-7) is below the pow/exp noise, so this is plain parity with such colours; that the PRODUCT applies the
    round trip is pinned bit for bit on the host build (tests/test_host_harness.py::test_percent_f_round_trip_is_applied)."""
    import dataclasses
    w, h = 320, 180
    sc = scenes.build_scene("default", w, h, 3)
    sc2 = dataclasses.replace(sc, defines=tuple(sc.defines[:9]) + (0.1234567, 0.05000004, 1e-7) + (0.3333333, 0.0999999, 0.2500001))
    ref, cnt = oracle.OracleScene(sc2, w, h, small_textures["textures"], small_textures["cubemap"]).render()
    ref_plain, _ = oracle.OracleScene(sc, w, h, small_textures["textures"], small_textures["cubemap"]).render()
    gl = wrapper.make_renderer(sc2, w, h, small_textures["textures"], small_textures["cubemap"])
    gl.draw()
    img = gl.read_pixels()
    gl.stop()
    assert float(np.abs(img - ref).max()) <= TOL
    assert float(np.abs(ref - ref_plain).max()) > 0.05


def test_specialize_refuses_depths_beyond_the_segment_cap_and_unbinds_cubemaps(small_textures):
    """rtx_specialize: iterations > 256 is an error, not a silent truncation; rtx_bind_texture(unit, 0) also clears the cube binding."""
    w, h = 96, 64
    sc = scenes.build_scene("default", w, h, 2)
    gl = wrapper.GLWrapper(w, h)
    assert gl.init_window()
    with pytest.raises(wrapper.RtxError, match="iterations"):
        gl.init_shaders(tuple(sc.defines[:8]) + (257,) + tuple(sc.defines[9:]))
    gl.init_shaders(tuple(sc.defines[:8]) + (256,) + tuple(sc.defines[9:]))
    gl.stop()
    gl = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"])
    gl.draw()
    with_sky = gl.read_pixels()
    gl.bind_texture(0, 0)                                       # glBindTexture(GL_TEXTURE_CUBE_MAP, 0)
    gl.draw()
    without = gl.read_pixels()
    gl.stop()
    ref, _ = oracle.OracleScene(sc, w, h, small_textures["textures"], None).render()
    assert float(np.abs(without - ref).max()) <= TOL and float(np.abs(with_sky - without).max()) > 0.01


def test_candidate_tables_follow_scene_updates(built, small_textures):
    """Scenes with long tables carry ray-pencil masks and slab tables that are derived from the scene: after update_buffer (a moved
    camera, a moved light, moved primitives -- what the reference's update_scene does every frame) the next draw must rebuild them.
    The frame of the updated context equals the frame of a fresh context given the same blocks, bit for bit; the option switches the
    tables off without changing a pixel; the stats report them."""
    import struct
    import random_scenes
    w, h = 160, 96
    first = random_scenes.crowd_scene(3, w, h)
    gl = wrapper.make_renderer(first, w, h, small_textures["textures"], small_textures["cubemap"])
    gl.draw()
    a0 = gl.read_pixels(wrapper.RTX_RGBA32F).copy()
    st = gl.stats()
    assert st["pencils"] >= 2 and st["last_pencil_build_ms"] > 0.0
    # the same tables, everything moved: camera 6 units to the side and turned, the light to the other side, every quadric / torus shifted
    blocks = dict(first.blocks)
    sb = bytearray(blocks["scene_buf"])
    sb[16:28] = struct.pack("<3f", 6.0, 1.5, -7.0)
    blocks["scene_buf"] = bytes(sb)
    lb = bytearray(blocks["lights_point_buf"])
    lb[0:12] = struct.pack("<3f", -9.0, 2.0, 25.0)
    blocks["lights_point_buf"] = bytes(lb)
    for name, rec, pos in (("surfaces_buf", 160, 112), ("toruses_buf", 112, 80)):
        buf = bytearray(blocks.get(name, b""))
        for k in range(len(buf) // rec):
            x, y, z = struct.unpack_from("<3f", buf, k * rec + pos)
            struct.pack_into("<3f", buf, k * rec + pos, x + 1.5, y - 0.7, z + 2.0)
            if name == "surfaces_buf":   # the clip box moves with the quadric
                for off in (80, 96):
                    v = struct.unpack_from("<3f", buf, k * rec + off)
                    struct.pack_into("<3f", buf, k * rec + off, *[c + d if abs(c) < 1e30 else c for c, d in zip(v, (1.5, -0.7, 2.0))])
        blocks[name] = bytes(buf)
    second = type(first)(blocks=blocks, defines=first.defines)
    gl.uploader.update(second)
    gl.draw()
    a1 = gl.read_pixels(wrapper.RTX_RGBA32F).copy()
    gl.set_option(wrapper.RTX_OPT_RAY_PENCILS, 0)
    gl.draw()
    a2 = gl.read_pixels(wrapper.RTX_RGBA32F).copy()
    assert gl.stats()["pencils"] == 0
    gl.stop()
    fresh = wrapper.make_renderer(second, w, h, small_textures["textures"], small_textures["cubemap"])
    fresh.set_option(wrapper.RTX_OPT_CULL, 0)
    fresh.draw()
    b = fresh.read_pixels(wrapper.RTX_RGBA32F)
    fresh.stop()
    assert (a0.view(np.uint32) != a1.view(np.uint32)).any(-1).mean() > 0.2          # the scene did change
    assert np.array_equal(a1.view(np.uint32), b.view(np.uint32))                   # updated context, tables rebuilt == literal scans of a fresh one
    assert np.array_equal(a2.view(np.uint32), b.view(np.uint32))                   # tables off


def test_stats_say_which_kernel_and_candidate_tables_a_scene_got(small_textures):
    """rtx_stats.kernel_variant / candidate_tables: no silent cliffs -- a scene with ray pencils always runs the many-primitive build
    (also below 32 primitives in total), the default scene the default one without tables, and switching the pencils off shows."""
    import random_scenes
    w, h = 160, 96
    gl = wrapper.make_renderer(scenes.build_scene("default", w, h, 2), w, h, small_textures["textures"], small_textures["cubemap"])
    gl.draw()
    st = gl.stats()
    assert (st["kernel_variant"], st["candidate_tables"], st["pencils"]) == (0, 0, 0)
    gl.stop()
    seen = set()
    for seed in range(12):
        sc = random_scenes.crowd_scene(seed, w, h)
        gl = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"])
        gl.draw()
        st = gl.stats()
        n_long = max(sc.defines[2], sc.defines[4])
        if st["pencils"] > 0:
            assert st["kernel_variant"] == 1 and (st["candidate_tables"] & 6) == 6 and 16 <= n_long <= 128, (seed, st, sc.defines[:6])
            seen.add(sum(sc.defines[:6]) >= 32)
            gl.set_option(wrapper.RTX_OPT_RAY_PENCILS, 0)
            gl.draw()
            assert (gl.stats()["candidate_tables"] & 6) == 0
        gl.stop()
    assert seen, "no crowd scene with ray pencils among the seeds"
