"""Multi-device contexts behind the C ABI (rtx_create_multi; SURVEY section 8(e), BASELINE north_star "GLWrapper dispatch -> HIP launch +
RCCL tile gather"): the frame a group of N ranks assembles on its root must be BIT-IDENTICAL to the frame one device traces -- both colour
targets, ray counters summed over the ranks, across consecutive frames (the buffer sets alternate), scene updates in between and with the
SMAA resolve on top. A GPU box here has ONE device, so the band split, the transfer plumbing and the placement kernel are exercised with
RTX_GATHER_PEER_COPY and all ranks on device 0; the RCCL transport runs where at least two devices exist (skipped otherwise) and is checked
for its error behaviour everywhere."""
import numpy as np
import pytest

from raytracing_opengl_amd import scenes, smaa_tables, wrapper

pytestmark = pytest.mark.gpu


def _n_devices():
    import torch
    return torch.cuda.device_count()


def _frames(gl, sc_list):
    out = []
    for sc in sc_list:
        if sc is not None:
            gl.uploader.update(sc)
        gl.draw()
        out.append((gl.read_pixels(wrapper.RTX_RGBA32F), gl.read_pixels(wrapper.RTX_RGBA8), gl.stats()))
    return out


@pytest.mark.parametrize("ranks", [2, 3, 5])
@pytest.mark.parametrize("kind,w,h,depth", [("default", 640, 360, 4), ("torus", 333, 207, 6), ("quadric", 320, 100, 4)])
def test_peer_copy_group_reproduces_the_single_device_frame(small_textures, ranks, kind, w, h, depth):
    seq = [None, scenes.build_scene(kind, w, h, depth, time=3.0, delta=0.1, yaw=20.0), scenes.build_scene(kind, w, h, depth, time=6.5, delta=0.1, yaw=-15.0, pitch=4.0), None]
    sc0 = scenes.build_scene(kind, w, h, depth)
    single = wrapper.make_renderer(sc0, w, h, small_textures["textures"], small_textures["cubemap"])
    single.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    want = _frames(single, seq)
    single.stop()
    group = wrapper.make_renderer(sc0, w, h, small_textures["textures"], small_textures["cubemap"], devices=[0] * ranks, gather=wrapper.RTX_GATHER_PEER_COPY)
    group.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    got = _frames(group, seq)
    group.stop()
    for k, ((f32, u8, st), (g32, g8, gst)) in enumerate(zip(want, got)):
        assert np.array_equal(f32.view(np.uint32), g32.view(np.uint32)), (k, int((f32.view(np.uint32) != g32.view(np.uint32)).sum()))
        assert np.array_equal(u8, g8), k
        assert (st["rays_closest"], st["rays_shadow"]) == (gst["rays_closest"], gst["rays_shadow"]), k
        assert gst["last_gather_ms"] > 0.0


def test_group_with_smaa_and_target_selection(small_textures):
    w, h = 480, 272
    sc = scenes.build_scene("default", w, h, 4)
    tables = smaa_tables.area_table(), smaa_tables.search_table()
    single = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"])
    single.enable_SMAA("ULTRA")
    single.set_smaa_tables(*tables)
    single.draw()
    want8, want_screen = single.read_pixels(wrapper.RTX_RGBA8), single.read_pixels(wrapper.RTX_SCREEN_RGBA8)
    single.stop()
    group = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"], devices=[0, 0, 0, 0], gather=wrapper.RTX_GATHER_PEER_COPY)
    group.set_option(wrapper.RTX_OPT_GATHER_TARGETS, 2)          # only what the reference's framebuffer holds travels
    group.enable_SMAA("ULTRA")
    group.set_smaa_tables(*tables)
    for _ in range(3):
        group.draw()
    assert np.array_equal(group.read_pixels(wrapper.RTX_RGBA8), want8)
    assert np.array_equal(group.read_pixels(wrapper.RTX_SCREEN_RGBA8), want_screen)
    with pytest.raises(wrapper.RtxError, match="not gathered"):
        group.read_pixels(wrapper.RTX_RGBA32F)
    group.stop()


def test_one_rank_group_is_a_plain_context_and_bad_arguments(small_textures):
    w, h = 160, 96
    sc = scenes.build_scene("default", w, h, 2)
    a = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"])
    a.draw()
    want = a.read_pixels()
    a.stop()
    b = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"], devices=[0])
    b.draw()
    assert np.array_equal(b.read_pixels().view(np.uint32), want.view(np.uint32))
    b.stop()
    gl = wrapper.GLWrapper(w, h, devices=[0, 0])                 # RCCL needs one device per rank
    assert not gl.init_window() and "listed twice" in gl.last_error
    gl = wrapper.GLWrapper(w, h, devices=[0, 99], gather=wrapper.RTX_GATHER_PEER_COPY)
    assert not gl.init_window() and "out of range" in gl.last_error


@pytest.mark.parametrize("ranks", [2, 4, 8])
def test_rccl_group_reproduces_the_single_device_frame(small_textures, ranks):
    """The RCCL transport proper: needs `ranks` devices (the driver's 8-GPU node; skipped on the single-GPU boxes)."""
    if _n_devices() < ranks:
        pytest.skip(f"{ranks} devices needed, {_n_devices()} present")
    w, h, depth = 1280, 720, 4
    seq = [None, scenes.build_scene("default", w, h, depth, time=2.0, delta=0.1), None]
    sc0 = scenes.build_scene("default", w, h, depth)
    single = wrapper.make_renderer(sc0, w, h, small_textures["textures"], small_textures["cubemap"])
    want = _frames(single, seq)
    single.stop()
    group = wrapper.make_renderer(sc0, w, h, small_textures["textures"], small_textures["cubemap"], devices=list(range(ranks)), gather=wrapper.RTX_GATHER_RCCL)
    got = _frames(group, seq)
    group.stop()
    for (f32, u8, _), (g32, g8, _g) in zip(want, got):
        assert np.array_equal(f32.view(np.uint32), g32.view(np.uint32)) and np.array_equal(u8, g8)
