"""Parity tests proper: the HIP tracer, called through the C ABI, against the oracle.

Bar (BASELINE.json north_star): max abs pixel difference <= 1e-4 on the float32 RGBA target,
and NaN pixels (the reference's NaN traps) in the same places. Ray counts -- a deterministic
function of scene, size and depth -- must be EXACTLY the oracle's.
"""
import numpy as np
import pytest

from oracle import oracle
from raytracing_opengl_amd import scenes, wrapper

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _render_gpu(sc, w, h, tex, opts=None):
    gl = wrapper.make_renderer(sc, w, h, tex["textures"], tex["cubemap"], device=0)
    gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    for k, v in (opts or {}).items():
        gl.set_option(k, v)
    gl.draw()
    img = gl.read_pixels(wrapper.RTX_RGBA32F)
    img8 = gl.read_pixels(wrapper.RTX_RGBA8)
    st = gl.stats()
    gl.stop()
    return img, img8, st


def _compare(img, ref):
    assert img.shape == ref.shape
    nan_mismatch = int((np.isnan(img) != np.isnan(ref)).sum())
    diff = np.abs(img - ref)
    diff = np.where(np.isnan(diff), 0.0, diff)
    return float(diff.max()), int((diff > TOL).sum()), nan_mismatch


def test_selftest_unorm8():
    gl = wrapper.GLWrapper(64, 64)
    assert gl.init_window(), getattr(gl, "last_error", "")
    assert gl.selftest() == 0
    gl.stop()


# BASELINE.json configs at sizes the oracle finishes in seconds
CASES = [
    ("default", 640, 480, 1),   # configs[0] at its own size
    ("default", 960, 540, 4),   # configs[1] scene/depth
    ("quadric", 480, 270, 4),   # configs[2] scene/depth
    ("torus", 320, 180, 6),     # configs[3] scene/depth
]


@pytest.mark.parametrize("lod", [1, 0])
@pytest.mark.parametrize("kind,w,h,depth", CASES)
def test_frame_parity_and_ray_counts(mid_textures, kind, w, h, depth, lod):
    sc = scenes.build_scene(kind, w, h, depth)
    ref, cnt = oracle.OracleScene(sc, w, h, mid_textures["textures"], mid_textures["cubemap"], texture_lod=lod).render()
    img, img8, st = _render_gpu(sc, w, h, mid_textures, {wrapper.RTX_OPT_TEXTURE_LOD: lod})
    mx, nbad, nanbad = _compare(img, ref)
    assert nanbad == 0
    assert mx <= TOL and nbad == 0, f"max diff {mx}, {nbad} components over {TOL}"
    assert st["rays_closest"] == cnt["rays_closest"]
    assert st["rays_shadow"] == cnt["rays_shadow"]
    # RGBA8 target = clamp + round of the same colours
    exp8 = (np.clip(np.nan_to_num(ref, nan=0.0), 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    assert np.abs(img8.astype(np.int16) - exp8.astype(np.int16)).max() <= 1
    # the PRODUCT variant (no ray counters: the kernel bench.py times) against the oracle directly, not only via "== counting variant"
    prod, _p8, _st = _render_gpu(sc, w, h, mid_textures, {wrapper.RTX_OPT_TEXTURE_LOD: lod, wrapper.RTX_OPT_COUNT_RAYS: 0})
    mx, nbad, nanbad = _compare(prod, ref)
    assert nanbad == 0 and mx <= TOL and nbad == 0, f"product variant: max diff {mx}, {nbad} components over {TOL}"
    assert np.array_equal(prod.view(np.uint32), img.view(np.uint32))


@pytest.mark.parametrize("opts", [{wrapper.RTX_OPT_CULL: 0}, {wrapper.RTX_OPT_SCENE_LDS: 1}, {wrapper.RTX_OPT_CULL: 0, wrapper.RTX_OPT_SCENE_LDS: 1}])
@pytest.mark.parametrize("kind,w,h,depth", [("default", 480, 270, 4), ("torus", 160, 90, 6), ("quadric", 240, 136, 4)])
def test_kernel_variants_agree_with_oracle(small_textures, kind, w, h, depth, opts):
    sc = scenes.build_scene(kind, w, h, depth)
    ref, cnt = oracle.OracleScene(sc, w, h, small_textures["textures"], small_textures["cubemap"]).render()
    img, _img8, st = _render_gpu(sc, w, h, small_textures, opts)
    mx, nbad, nanbad = _compare(img, ref)
    assert nanbad == 0 and mx <= TOL and nbad == 0
    assert st["rays_closest"] == cnt["rays_closest"] and st["rays_shadow"] == cnt["rays_shadow"]


# ---- texture rule phase B: mip chain + trilinear + quad-derivative LOD ----
LOD_CASES = [("default", 960, 540, 4, {}), ("default", 640, 360, 5, dict(time=4.0, delta=0.3, yaw=55.0, pitch=-4.0, cam_pos=(2.0, 1.0, -4.0))),
             ("default", 333, 207, 3, {}),  # odd framebuffer: helper invocations complete the edge quads
             ("default", 640, 360, 4, dict(yaw=80.0, pitch=3.0, cam_pos=(0.0, 0.5, 2.0)))]  # looking at the crate / Saturn


@pytest.mark.parametrize("kind,w,h,depth,kw", LOD_CASES)
def test_frame_parity_with_mip_lod(mid_textures, kind, w, h, depth, kw):
    sc = scenes.build_scene(kind, w, h, depth, **kw)
    ref, cnt = oracle.OracleScene(sc, w, h, mid_textures["textures"], mid_textures["cubemap"], texture_lod=1).render()
    ref0, _ = oracle.OracleScene(sc, w, h, mid_textures["textures"], mid_textures["cubemap"], texture_lod=0).render()
    img, _img8, st = _render_gpu(sc, w, h, mid_textures, {wrapper.RTX_OPT_TEXTURE_LOD: 1})
    mx, nbad, nanbad = _compare(img, ref)
    assert nanbad == 0 and mx <= TOL and nbad == 0, f"max diff {mx}, {nbad} components over {TOL}"
    assert st["rays_closest"] == cnt["rays_closest"] and st["rays_shadow"] == cnt["rays_shadow"]
    assert np.abs(ref - ref0).max() > 1e-3  # the LOD rule actually changes pixels on this view


def test_lod_bands_match_full_frame(small_textures):
    """Quads never straddle row bands (multiples of 8 rows), so LOD results are band-independent."""
    import torch
    from raytracing_opengl_amd import bands
    w, h, world, band_rows = 320, 200, 3, 8
    sc = scenes.build_scene("default", w, h, 4)
    gl = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"], texture_lod=1)
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


@pytest.mark.parametrize("lod", [1, 0])
@pytest.mark.parametrize("name", ["glass_tir", "inside_box", "degenerate_rings"])
def test_trap_scenes_on_gpu(small_textures, name, lod):
    """TIR, refractive boxes with the shared 256-segment cap, origins inside boxes (T21), exact-zero
    direction components (T5), degenerate quadric branch (T4), two textured rings in one shadow ray (T10)."""
    import trap_scenes
    w, h = (321, 181) if name == "inside_box" else (320, 180)
    sc = trap_scenes.ALL[name](w, h)
    ref, cnt = oracle.OracleScene(sc, w, h, small_textures["textures"], small_textures["cubemap"], texture_lod=lod).render()
    img, _i8, st = _render_gpu(sc, w, h, small_textures, {wrapper.RTX_OPT_TEXTURE_LOD: lod})
    mx, nbad, nanbad = _compare(img, ref)
    assert nanbad == 0 and mx <= TOL and nbad == 0, f"{name}: max diff {mx}, {nbad} over"
    assert st["rays_closest"] == cnt["rays_closest"] and st["rays_shadow"] == cnt["rays_shadow"]


@pytest.mark.parametrize("w,h", [(960, 540), (333, 207), (1000, 40)])
def test_xcd_remap_is_a_pure_reordering(small_textures, w, h):
    """The XCD-aware super-tile workgroup order must give the bit-identical frame and ray counts."""
    sc = scenes.build_scene("default", w, h, 4)
    a, _a8, sa = _render_gpu(sc, w, h, small_textures, {wrapper.RTX_OPT_XCD_REMAP: 1})
    b, _b8, sb = _render_gpu(sc, w, h, small_textures, {wrapper.RTX_OPT_XCD_REMAP: 0})
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert (sa["rays_closest"], sa["rays_shadow"]) == (sb["rays_closest"], sb["rays_shadow"])


def test_moving_camera_and_animation(small_textures):
    """Per-frame update_buffer path (reference SceneManager.cpp:257-276): same context, new blocks."""
    w, h = 320, 180
    sc0 = scenes.build_scene("default", w, h, 5)
    gl = wrapper.make_renderer(sc0, w, h, small_textures["textures"], small_textures["cubemap"])
    for t, yaw, pitch, pos in [(0.0, 0.0, 0.0, None), (3.0, 30.0, -8.0, (-4.0, 1.0, -6.0)), (9.5, -60.0, 15.0, (7.0, 3.0, -2.0))]:
        sc = scenes.build_scene("default", w, h, 5, time=t, delta=0.4 * t, yaw=yaw, pitch=pitch, cam_pos=pos)
        gl.uploader.update(sc)
        gl.draw()
        img = gl.read_pixels()
        ref, _ = oracle.OracleScene(sc, w, h, small_textures["textures"], small_textures["cubemap"]).render()
        mx, nbad, nanbad = _compare(img, ref)
        assert nanbad == 0 and mx <= TOL, (t, mx)
    gl.stop()


def test_odd_sizes_and_window_not_multiple_of_tile(small_textures):
    w, h = 333, 207  # framebuffer odd; canvas bumped to even like reference main.cpp:39-41
    sc = scenes.build_scene("default", w, h, 3)
    ref, _ = oracle.OracleScene(sc, w, h, small_textures["textures"], small_textures["cubemap"]).render()
    img, _i8, _st = _render_gpu(sc, w, h, small_textures)
    mx, nbad, nanbad = _compare(img, ref)
    assert nanbad == 0 and mx <= TOL


def test_row_bands_reassemble_the_frame(small_textures):
    """rtx_draw_bands: interleaved 16-row bands of 3 'ranks' == the full frame, bit for bit."""
    import torch
    from raytracing_opengl_amd import bands
    w, h, world, band_rows = 320, 200, 3, 16
    sc = scenes.build_scene("default", w, h, 4)
    gl = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"])
    gl.draw()
    full = torch.from_numpy(gl.read_pixels())
    parts = []
    rows_max = bands.max_local_rows(h, band_rows, world)
    for r in range(world):
        buf = torch.zeros((rows_max, w, 4), dtype=torch.float32, device="cuda:0")
        gl.draw_bands(band_rows, r, world, buf.data_ptr(), wrapper.RTX_RGBA32F)
        gl.finish()
        parts.append(buf.cpu())
    frame = bands.unpermute(parts, h, band_rows, world)
    assert torch.equal(frame.view(torch.int32), full.view(torch.int32))
    gl.stop()


def test_contiguous_row_ranges_reassemble_the_frame(small_textures):
    """rtx_draw_rows: one rank's share in the contiguous band layout -- ranges starting on multiples of 8, of any length incl. the odd rest of
    the frame and zero rows -- packed from the start of the destination; together the ranges are the frame, bit for bit."""
    import torch
    w, h = 333, 207
    sc = scenes.build_scene("default", w, h, 4)
    gl = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"])
    gl.draw()
    want = gl.read_pixels(wrapper.RTX_RGBA32F)
    stream = torch.cuda.current_stream().cuda_stream
    buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    got = np.zeros_like(want)
    y = 0
    for n in (8, 0, 64, 72, 40, 23):
        gl.draw_rows(y, n, buf.data_ptr(), wrapper.RTX_RGBA32F, stream)
        gl.finish()
        got[y:y + n] = buf[:n].cpu().numpy()
        y += n
    assert y == h and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    with pytest.raises(wrapper.RtxError, match="multiple of 8"):
        gl.draw_rows(4, 8, buf.data_ptr(), wrapper.RTX_RGBA32F, stream)
    with pytest.raises(wrapper.RtxError):
        gl.draw_rows(200, 16, buf.data_ptr(), wrapper.RTX_RGBA32F, stream)
    gl.stop()


def test_high_occupancy_variant_is_bit_identical(small_textures):
    """RTX_OPT_HIGH_OCCUPANCY selects another register budget of the same kernel (7 waves/SIMD instead of 6;
    auto-selected for scenes with >= 32 primitives): the frames must not differ in a single bit."""
    w, h = 320, 184
    for name in ("default", "quadric"):
        sc = scenes.build_scene(name, w, h, 4)
        gl = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"])
        frames = []
        for mode in (0, 1, -1):
            gl.set_option(wrapper.RTX_OPT_HIGH_OCCUPANCY, mode)
            gl.draw()
            frames.append(gl.read_pixels().copy())
        gl.stop()
        assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32)), name
        assert np.array_equal(frames[0].view(np.uint32), frames[2].view(np.uint32)), name


def test_rgba8_row_bands_reassemble_the_rgba8_frame(small_textures):
    """The multi-GPU bench gathers the RGBA8 target: banded RGBA8 draws == the full RGBA8 frame, byte for byte."""
    import torch
    from raytracing_opengl_amd import bands
    w, h, world, band_rows = 328, 200, 4, 8
    sc = scenes.build_scene("default", w, h, 4)
    gl = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"])
    gl.draw()
    full = torch.from_numpy(gl.read_pixels(wrapper.RTX_RGBA8))
    parts = []
    rows_max = bands.max_local_rows(h, band_rows, world)
    for r in range(world):
        buf = torch.zeros((rows_max, w, 4), dtype=torch.uint8, device="cuda:0")
        gl.draw_bands(band_rows, r, world, buf.data_ptr(), wrapper.RTX_RGBA8)
        gl.finish()
        parts.append(buf.cpu())
    frame = bands.unpermute(parts, h, band_rows, world)
    assert torch.equal(frame, full.view(h, w, 4))
    gl.stop()


def test_full_size_properties(mid_textures):
    """BASELINE size (3840x2160, depth 4): size-independent properties instead of a full oracle frame:
    (1) oracle parity on a set of rows sampled across the frame, (2) exact ray count on those rows is
    covered by the band draw + counter, (3) determinism: two draws are bit-identical."""
    w, h, depth = 3840, 2160, 4
    sc = scenes.build_scene("default", w, h, depth)
    gl = wrapper.make_renderer(sc, w, h, mid_textures["textures"], mid_textures["cubemap"])
    gl.draw()
    a = gl.read_pixels()
    gl.draw()
    b = gl.read_pixels()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    o = oracle.OracleScene(sc, w, h, mid_textures["textures"], mid_textures["cubemap"])
    for y0 in (0, 536, 1000, 1080, 1304, 2152):
        ref, _ = o.render(y0, y0 + 8)
        mx, nbad, nanbad = _compare(a[y0:y0 + 8], ref)
        assert nanbad == 0 and mx <= TOL, (y0, mx)
    assert np.all(a[..., 3] == 1.0)
    gl.stop()


def test_config1_at_its_own_size(mid_textures):
    """BASELINE configs[1] -- default scene, 1920 x 1080, depth 4, one GPU -- at exactly that size: the WHOLE frame of the product
    variant against the oracle (the oracle renders 1080p in seconds), exact ray counts from the counting variant, and the two variants
    bit-identical."""
    w, h, depth = 1920, 1080, 4
    sc = scenes.build_scene("default", w, h, depth)
    gl = wrapper.make_renderer(sc, w, h, mid_textures["textures"], mid_textures["cubemap"])
    gl.draw()
    a = gl.read_pixels()
    ref, cnt = oracle.OracleScene(sc, w, h, mid_textures["textures"], mid_textures["cubemap"]).render()
    mx, nbad, nanbad = _compare(a, ref)
    assert nanbad == 0 and mx <= TOL, mx
    gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    gl.draw()
    st = gl.stats()
    assert (st["rays_closest"], st["rays_shadow"]) == (cnt["rays_closest"], cnt["rays_shadow"])
    assert np.array_equal(gl.read_pixels().view(np.uint32), a.view(np.uint32))
    gl.stop()


@pytest.mark.parametrize("kind,depth,rows", [("quadric", 4, (0, 640, 1080, 1504, 2152)), ("torus", 6, (320, 1080, 1400))])
def test_full_size_stress_configs(small_textures, kind, depth, rows):
    """BASELINE.json configs[2] / configs[3] at their own size (3840x2160): two draws bit-identical, oracle
    parity and exact ray counts on 8-row bands sampled across the frame (the oracle needs minutes for a
    whole frame of these scenes)."""
    import torch
    w, h = 3840, 2160
    sc = scenes.build_scene(kind, w, h, depth)
    gl = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"])
    gl.draw()
    a = gl.read_pixels()
    gl.draw()
    b = gl.read_pixels()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    o = oracle.OracleScene(sc, w, h, small_textures["textures"], small_textures["cubemap"])
    gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    band = torch.zeros((8, w, 4), dtype=torch.float32, device="cuda:0")
    for y0 in rows:
        ref, cnt = o.render(y0, y0 + 8)
        mx, nbad, nanbad = _compare(a[y0:y0 + 8], ref)
        assert nanbad == 0 and mx <= TOL, (kind, y0, mx)
        gl.draw_bands(8, y0 // 8, 1 << 20, band.data_ptr(), wrapper.RTX_RGBA32F)  # exactly the band starting at y0
        gl.finish()
        st = gl.stats()
        assert (st["rays_closest"], st["rays_shadow"]) == (cnt["rays_closest"], cnt["rays_shadow"]), (kind, y0)
        assert np.array_equal(band.cpu().numpy().view(np.uint32), a[y0:y0 + 8].view(np.uint32))
    gl.stop()


def test_config4_8k_frame_from_eight_band_shares(small_textures):
    """BASELINE.json configs[4] (default scene, 7680x4320, depth 4, 8 ranks) on one GPU: each of the 8 ranks' interleaved
    8-row band shares is traced exactly as bench.py traces it (RGBA8 target), the root's un-permute gives the full RGBA8
    frame byte for byte, the shares' ray counts add up to the frame's, and the float frame matches the oracle on rows
    sampled across the image."""
    import torch
    from raytracing_opengl_amd import bands
    w, h, depth, world, band_rows = 7680, 4320, 4, 8, 8
    sc = scenes.build_scene("default", w, h, depth)
    gl = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"])
    gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    gl.draw()
    gl.finish()
    st = gl.stats()
    total = (st["rays_closest"], st["rays_shadow"])
    full8 = torch.from_numpy(gl.read_pixels(wrapper.RTX_RGBA8)).view(h, w, 4)
    full32 = gl.read_pixels()
    rows_max = bands.max_local_rows(h, band_rows, world)
    parts, got = [], [0, 0]
    for r in range(world):
        buf = torch.zeros((rows_max, w, 4), dtype=torch.uint8, device="cuda:0")
        gl.draw_bands(band_rows, r, world, buf.data_ptr(), wrapper.RTX_RGBA8)
        gl.finish()
        st = gl.stats()
        got[0] += st["rays_closest"]
        got[1] += st["rays_shadow"]
        parts.append(buf.cpu())
    assert tuple(got) == total
    assert torch.equal(bands.unpermute(parts, h, band_rows, world), full8)
    o = oracle.OracleScene(sc, w, h, small_textures["textures"], small_textures["cubemap"])
    for y0 in (0, 1072, 2000, 2160, 2608, 4312):
        ref, _ = o.render(y0, y0 + 8)
        mx, nbad, nanbad = _compare(full32[y0:y0 + 8], ref)
        assert nanbad == 0 and mx <= TOL, (y0, mx)
    gl.stop()


def _golden_ids():
    import golden_frames
    import os
    return golden_frames.FILES, [os.path.basename(p)[:-4] for p in golden_frames.FILES]


@pytest.mark.parametrize("lod", [1, 0])
@pytest.mark.parametrize("path", _golden_ids()[0], ids=_golden_ids()[1])
def test_committed_golden_frames(path, lod):
    """The HIP tracer against the committed golden vectors (inputs + expected frames in one .npz)."""
    import golden_frames
    g = golden_frames.load(path)
    tex = {"textures": g["textures"], "cubemap": g["cubemap"]}
    img, _i8, st = _render_gpu(g["scene"], g["width"], g["height"], tex, {wrapper.RTX_OPT_TEXTURE_LOD: lod})
    mx, nbad, nanbad = _compare(img, g["frames"][lod])
    assert nanbad == 0 and mx <= TOL and nbad == 0
    assert (st["rays_closest"], st["rays_shadow"]) == g["rays"][lod]


def test_error_behaviour():
    """Order and name errors mirror the reference's failure points (GLWrapper.cpp:360,370-375)."""
    gl = wrapper.GLWrapper(64, 64)
    assert gl.init_window()
    with pytest.raises(wrapper.RtxError):
        gl.init_buffer("spheres_buf", 1, b"\0" * 112)  # before init_shaders
    gl.init_shaders((0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0))
    with pytest.raises(wrapper.RtxError, match="Invalid ubo block name"):
        gl.init_buffer("no_such_buf", 0, b"")
    with pytest.raises(wrapper.RtxError):
        gl.draw()  # blocks missing
    # load_cubemap(faces, genMipmap): both forms are accepted (cube mips: tests/test_gpu_cube_mips.py); bad face sizes are not
    import numpy as np
    faces = [np.full((8, 8, 3), 40 * k, np.uint8) for k in range(6)]
    assert gl.load_cubemap(faces, True) > 0
    assert gl.load_cubemap(faces, False) > 0
    gl.stop()


@pytest.mark.parametrize("kind,w,h,depth", [("default", 3840, 2160, 4), ("quadric", 3840, 2160, 4), ("torus", 3840, 2160, 6), ("default", 7680, 4320, 4)])
def test_culls_and_kernel_variants_change_nothing_at_full_size(kind, w, h, depth):
    """Every BASELINE configuration at its FULL size: the product path (culls on; the many-primitive variant with its group culls where it
    is selected) against the literal scans (RTX_OPT_CULL = 0) on the GPU itself -- frames bit for bit, ray counters equal. The oracle
    cannot audit 8 M pixels per case in seconds; the un-culled kernel, itself pinned to the oracle at small sizes above, can."""
    from raytracing_opengl_amd import textures
    ts = textures.default_texture_set(scale=4)
    sc = scenes.build_scene(kind, w, h, depth)
    frames, counts, pencils = [], [], []
    for cull, count, pen in ((1, 0, 1), (1, 0, 0), (1, 1, 1), (0, 1, 1)):   # product variant with / without ray pencils, counting variant, literal scans
        gl = wrapper.make_renderer(sc, w, h, ts["textures"], ts["cubemap"])
        gl.set_option(wrapper.RTX_OPT_CULL, cull)
        gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, count)
        gl.set_option(wrapper.RTX_OPT_RAY_PENCILS, pen)
        gl.draw()
        frames.append(gl.read_pixels(wrapper.RTX_RGBA32F).view(np.uint32))
        st = gl.stats()
        counts.append((st["rays_closest"], st["rays_shadow"]))
        pencils.append(st["pencils"])
        gl.stop()
    assert all(np.array_equal(f, frames[3]) for f in frames[:3])
    assert pencils[0] == (0 if kind == "default" else 3) and pencils[1] == 0      # camera, point light, directional light
    assert counts[2] == counts[3] and counts[2][0] > w * h
