"""GLWrapper::load_cubemap(faces, genMipmap = true) on the GPU (reference GLWrapper.cpp:307-310: glGenerateMipmap(GL_TEXTURE_CUBE_MAP) and
GL_LINEAR_MIPMAP_LINEAR, so that texture(skybox, rd), rt.frag:893, is trilinear): the HIP kernels' SKYLOD instantiations, through the C ABI,
against the oracle's rule (cube part) at the north star's 1e-4 with equal ray counts; against the reference's own
shader on llvmpipe: tests/test_reference_frames.py (GPU_PLAN, the *_cube_mips fixtures)."""
import numpy as np
import pytest

from oracle import oracle
from raytracing_opengl_amd import scenes, wrapper

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _gpu(sc, w, h, tex, opts=None, cube_mipmap=True):
    gl = wrapper.make_renderer(sc, w, h, tex["textures"], tex["cubemap"], cube_mipmap=cube_mipmap)
    gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    for k, v in (opts or {}).items():
        gl.set_option(k, v)
    gl.draw()
    img = gl.read_pixels(wrapper.RTX_RGBA32F)
    st = gl.stats()
    gl.stop()
    return img, st


def _judge(img, ref):
    assert int((np.isnan(img) != np.isnan(ref)).sum()) == 0
    d = np.abs(img - ref)
    d = np.where(np.isnan(d), 0.0, d)
    assert float(d.max()) <= TOL, (float(d.max()), int((d > TOL).sum()))


# the default scene from far away (the sky is minified: lambda 1 - 3 at these sizes with 512-texel faces), from its usual pose, an odd
# framebuffer (helper invocations complete the edge quads), and the two many-primitive scenes (their own kernel instantiation)
CASES = [("default", 480, 270, 4, {}), ("default", 333, 207, 3, dict(time=4.0, delta=0.3, yaw=55.0, pitch=-4.0, cam_pos=(2.0, 1.0, -4.0))),
         ("default", 160, 90, 5, dict(yaw=-120.0, pitch=35.0)), ("quadric", 240, 136, 4, {}), ("torus", 160, 90, 6, {})]


@pytest.mark.parametrize("kind,w,h,depth,kw", CASES)
def test_frame_parity_with_a_mip_mapped_sky_box(mid_textures, kind, w, h, depth, kw):
    sc = scenes.build_scene(kind, w, h, depth, **kw)
    args = (sc, w, h, mid_textures["textures"], mid_textures["cubemap"])
    ref, cnt = oracle.OracleScene(*args, texture_lod=1, cube_mipmap=True).render()
    flat, _ = oracle.OracleScene(*args, texture_lod=1, cube_mipmap=False).render()
    assert np.abs(ref - flat).max() > 1e-2                       # the cube mips change pixels of this view
    img, st = _gpu(sc, w, h, mid_textures)
    _judge(img, ref)
    assert st["rays_closest"] == cnt["rays_closest"] and st["rays_shadow"] == cnt["rays_shadow"]
    # the product variant (no ray counters) and the literal scans (no culls): the same bits
    for opts in ({wrapper.RTX_OPT_COUNT_RAYS: 0}, {wrapper.RTX_OPT_CULL: 0}, {wrapper.RTX_OPT_CULL: 0, wrapper.RTX_OPT_COUNT_RAYS: 0}):
        other, _ = _gpu(sc, w, h, mid_textures, opts)
        assert np.array_equal(other.view(np.uint32), img.view(np.uint32)), opts


def test_the_flag_off_and_lod_off_sample_level_zero(small_textures):
    """genMipmap = false (the reference's default) and RTX_OPT_TEXTURE_LOD = 0 both leave the sky at level 0: the same bits as before the
    feature existed, and as each other."""
    w, h = 320, 180
    sc = scenes.build_scene("default", w, h, 4)
    args = (sc, w, h, small_textures["textures"], small_textures["cubemap"])
    for lod in (1, 0):
        ref, _ = oracle.OracleScene(*args, texture_lod=lod, cube_mipmap=False).render()
        a, _ = _gpu(sc, w, h, small_textures, {wrapper.RTX_OPT_TEXTURE_LOD: lod}, cube_mipmap=False)
        _judge(a, ref)
    ref0, _ = oracle.OracleScene(*args, texture_lod=0, cube_mipmap=True).render()
    b, _ = _gpu(sc, w, h, small_textures, {wrapper.RTX_OPT_TEXTURE_LOD: 0}, cube_mipmap=True)
    c, _ = _gpu(sc, w, h, small_textures, {wrapper.RTX_OPT_TEXTURE_LOD: 0}, cube_mipmap=False)
    _judge(b, ref0)
    assert np.array_equal(b.view(np.uint32), c.view(np.uint32))


def test_missing_faces_odd_face_sizes_and_one_texel_faces(small_textures):
    """A face that failed to load (GLWrapper.cpp:296-305 skips it) leaves a mip-mapped cube map incomplete: glGenerateMipmap raises
    GL_INVALID_OPERATION and every face samples (0, 0, 0, 1) -- the frame is the frame of six black faces; a face size that is
    not a power of two (levels of max(1, n >> L) texels); 1 x 1 faces (a chain of one level)."""
    w, h = 200, 120
    sc = scenes.build_scene("default", w, h, 3, yaw=-120.0, pitch=35.0)
    rng = np.random.default_rng(5)
    for n, missing in ((37, (1, 4)), (1, ()), (96, (0,))):
        faces = [None if f in missing else rng.integers(0, 256, (n, n, 3), dtype=np.uint8) for f in range(6)]
        tex = dict(textures=small_textures["textures"], cubemap=faces)
        ref, cnt = oracle.OracleScene(sc, w, h, tex["textures"], faces, texture_lod=1, cube_mipmap=True).render()
        img, st = _gpu(sc, w, h, tex)
        _judge(img, ref)
        assert st["rays_closest"] == cnt["rays_closest"]
        if missing:
            black = dict(textures=small_textures["textures"], cubemap=[np.zeros((n, n, 3), np.uint8) for _ in range(6)])
            img_black, _ = _gpu(sc, w, h, black)
            assert np.array_equal(img.view(np.uint32), img_black.view(np.uint32)), "an incomplete mip-mapped cube map samples black on every face"
            # without mips the missing face alone is black (rounds 1-5): the other faces still show
            img0, _ = _gpu(sc, w, h, tex, cube_mipmap=False)
            assert not np.array_equal(img0.view(np.uint32), img_black.view(np.uint32))


def test_row_bands_of_a_mip_mapped_sky_equal_the_full_frame(small_textures):
    """Quads never straddle the 8-row bands of the multi-GPU split: band-wise draws give the full frame's bits."""
    import torch
    from raytracing_opengl_amd import bands
    w, h, world, band_rows = 320, 200, 3, 8
    sc = scenes.build_scene("default", w, h, 4, yaw=-120.0, pitch=35.0)
    gl = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"], cube_mipmap=True)
    gl.draw()
    full = torch.from_numpy(gl.read_pixels())
    parts = []
    for r in range(world):
        buf = torch.zeros((bands.max_local_rows(h, band_rows, world), w, 4), dtype=torch.float32, device="cuda:0")
        gl.draw_bands(band_rows, r, world, buf.data_ptr(), wrapper.RTX_RGBA32F)
        gl.finish()
        parts.append(buf.cpu())
    assert torch.equal(bands.unpermute(parts, h, band_rows, world).view(torch.int32), full.view(torch.int32))
    gl.stop()


def test_scene_in_lds_with_cube_mips_is_refused_loudly(small_textures):
    """The LDS-staged experiment has no SKYLOD instantiation: rtx_draw says so instead of sampling level 0 silently."""
    sc = scenes.build_scene("default", 64, 64, 2)
    gl = wrapper.make_renderer(sc, 64, 64, small_textures["textures"], small_textures["cubemap"], cube_mipmap=True)
    gl.set_option(wrapper.RTX_OPT_SCENE_LDS, 1)
    with pytest.raises(wrapper.RtxError, match="mip-mapped sky box"):
        gl.draw()
    gl.set_option(wrapper.RTX_OPT_SCENE_LDS, 0)
    gl.draw()
    gl.stop()


@pytest.mark.parametrize("gen,seeds", [("random_scene", 40), ("nasty_scene", 30), ("scaled_quat_scene", 20)])
def test_fuzz_scenes_with_a_mip_mapped_sky_box(small_textures, gen, seeds):
    """Random content (tests/random_scenes.py: glass, mirrors, degenerate records, non-unit quaternions) under a mip-mapped sky box: the sky is
    reached at every bounce depth and in divergent quads; counting variant against the oracle (the fuzz bar of tests/parity_bar.py, equal ray
    counts), product variant bit for bit."""
    import parity_bar
    import random_scenes
    make = getattr(random_scenes, gen)
    bad, changed = [], 0
    for seed in range(31000, 31000 + seeds):
        w, h = ((96, 64), (97, 65))[seed % 2]
        sc = make(seed, w, h)
        args = (sc, w, h, small_textures["textures"], small_textures["cubemap"])
        ref, cnt = oracle.OracleScene(*args, texture_lod=1, cube_mipmap=True).render()
        flat, _ = oracle.OracleScene(*args, texture_lod=1, cube_mipmap=False).render()
        with np.errstate(invalid="ignore"):
            changed += bool(np.nanmax(np.abs(np.where(np.isfinite(ref) & np.isfinite(flat), ref - flat, 0.0))) > 1e-3)
        img, st = _gpu(sc, w, h, small_textures)
        prod, _ = _gpu(sc, w, h, small_textures, {wrapper.RTX_OPT_COUNT_RAYS: 0})
        ok = parity_bar.judge(img, ref)["ok"] and st["rays_closest"] == cnt["rays_closest"] and st["rays_shadow"] == cnt["rays_shadow"]
        ok = ok and np.array_equal(prod.view(np.uint32), img.view(np.uint32))
        if not ok:
            bad.append(seed)
    assert not bad, bad
    assert changed > seeds // 2, changed       # the cube mips are in these pictures
