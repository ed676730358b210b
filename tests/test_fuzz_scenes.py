"""Random scenes (tests/random_scenes.py): the product's device code must reproduce the oracle -- bit for bit on the host
build (level-0 textures, culls on and off), within 1e-4 with identical ray counts on the GPU (reference texture state)."""
import os

import numpy as np
import pytest

import harness
import parity_bar
import random_scenes
from oracle import oracle

@pytest.mark.parametrize("seed", list(range(24)) + [9029])
def test_random_scene_bit_exact_on_host(built, small_textures, seed):
    W, H = (323, 181) if seed == 9029 else [(112, 64), (113, 65)][seed % 2]   # 9029: a mirror ray straight down a torus tube (cull regression)
    sc = random_scenes.random_scene(seed, W, H)
    ref, cnt = oracle.OracleScene(sc, W, H, small_textures["textures"], small_textures["cubemap"], texture_lod=0).render()
    for cull in (True, False):
        img, hc = harness.render(sc, W, H, small_textures["textures"], small_textures["cubemap"], cull=cull)
        same = (img.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(img) & np.isnan(ref))
        assert same.all(), (seed, cull, int((~same).sum()))
        assert hc["closest"] == cnt["rays_closest"] and hc["shadow_ref"] == cnt["rays_shadow"], (seed, cull)


# 1566, 1785: overflowed mask x black mirror miss must give NaN like the shader; 3690: zero-tube torus, spurious solver root;
# 201072 (at 323x181): non-unit ray directions, for which the solver reports phantom roots -- torus culls must stand aside
@pytest.mark.parametrize("seed", list(range(16)) + [1566, 1785, 3690, 201072])
def test_nasty_scene_bit_exact_on_host(built, small_textures, seed):
    """Degenerate configurations on purpose (tests/random_scenes.py::nasty_scene)."""
    W, H = (323, 181) if seed == 201072 else [(97, 61), (96, 60), (65, 97), (121, 67)][seed % 4]
    sc = random_scenes.nasty_scene(seed, W, H)
    ref, cnt = oracle.OracleScene(sc, W, H, small_textures["textures"], small_textures["cubemap"], texture_lod=0).render()
    for cull in (True, False):
        img, hc = harness.render(sc, W, H, small_textures["textures"], small_textures["cubemap"], cull=cull)
        same = (img.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(img) & np.isnan(ref))
        assert same.all(), (seed, cull, int((~same).sum()))
        assert hc["closest"] == cnt["rays_closest"] and hc["shadow_ref"] == cnt["rays_shadow"], (seed, cull)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 7, 1566, 1785, 3690, 4001, 4002, 4003, 201072])
def test_nasty_scene_on_gpu(built, small_textures, seed):
    from raytracing_opengl_amd import wrapper
    w, h = (323, 181) if seed == 201072 else [(97, 61), (96, 60), (65, 97), (121, 67)][seed % 4]
    sc = random_scenes.nasty_scene(seed, w, h)
    ref, cnt = oracle.OracleScene(sc, w, h, small_textures["textures"], small_textures["cubemap"], texture_lod=1).render()
    gl = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"])
    gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    gl.draw()
    img = gl.read_pixels()
    st = gl.stats()
    gl.stop()
    fin = np.isfinite(img) & np.isfinite(ref)
    assert (np.isnan(img) == np.isnan(ref)).all() and (np.isinf(img) == np.isinf(ref)).all(), seed
    assert float(np.abs(np.where(fin, img - ref, 0.0)).max()) <= 1e-4, seed
    assert st["rays_closest"] == cnt["rays_closest"] and st["rays_shadow"] == cnt["rays_shadow"], seed


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(40))
def test_random_scene_on_gpu(built, small_textures, seed):
    from raytracing_opengl_amd import wrapper
    w, h = [(160, 96), (161, 97), (323, 181), (96, 160)][seed % 4]   # odd sizes run helper invocations; centre row/column rays are axis-parallel
    sc = random_scenes.random_scene(1000 + seed, w, h)
    ref, cnt = oracle.OracleScene(sc, w, h, small_textures["textures"], small_textures["cubemap"], texture_lod=1).render()
    gl = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"])
    gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    gl.draw()
    img = gl.read_pixels()
    st = gl.stats()
    gl.stop()
    both_nan = np.isnan(img) & np.isnan(ref)
    d = np.where(both_nan, 0.0, np.abs(img - ref))
    assert not (np.isnan(img) ^ np.isnan(ref)).any(), seed
    assert np.nanmax(d) <= 1e-4, (seed, float(np.nanmax(d)))
    assert st["rays_closest"] == cnt["rays_closest"] and st["rays_shadow"] == cnt["rays_shadow"], seed


@pytest.mark.parametrize("seed", range(24))
def test_scaled_quaternion_scene_bit_exact_on_host(built, small_textures, seed):
    """Rotation quaternions of norm != 1 (tests/random_scenes.py::scaled_quat_scene): culls on == culls off == oracle."""
    W, H = [(112, 64), (113, 65)][seed % 2]
    sc = random_scenes.scaled_quat_scene(seed, W, H)
    ref, cnt = oracle.OracleScene(sc, W, H, small_textures["textures"], small_textures["cubemap"], texture_lod=0).render()
    for cull in (True, False):
        img, hc = harness.render(sc, W, H, small_textures["textures"], small_textures["cubemap"], cull=cull)
        same = (img.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(img) & np.isnan(ref))
        assert same.all(), (seed, cull, int((~same).sum()))
        assert hc["closest"] == cnt["rays_closest"] and hc["shadow_ref"] == cnt["rays_shadow"], (seed, cull)


@pytest.mark.parametrize("seed", range(16))
def test_crowd_scene_bit_exact_on_host(built, small_textures, seed):
    """Long tables (tests/random_scenes.py::crowd_scene): the second-level group culls must not change a bit or a ray count."""
    W, H = [(96, 54), (97, 55)][seed % 2]
    sc = random_scenes.crowd_scene(seed, W, H)
    ref, cnt = oracle.OracleScene(sc, W, H, small_textures["textures"], small_textures["cubemap"], texture_lod=0).render()
    for cull in (True, False):
        img, hc = harness.render(sc, W, H, small_textures["textures"], small_textures["cubemap"], cull=cull)
        same = (img.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(img) & np.isnan(ref))
        assert same.all(), (seed, cull, int((~same).sum()))
        assert hc["closest"] == cnt["rays_closest"] and hc["shadow_ref"] == cnt["rays_shadow"], (seed, cull)


@pytest.mark.parametrize("seed", list(range(16)) + [966, 9038])
def test_pencil_scene_bit_exact_on_host(built, small_textures, seed):
    """crowd_scene's long tables under lights / cameras that stress the ray pencils (tests/random_scenes.py::pencil_scene): all culls incl.
    the pencils == two-level culls only == no culls == oracle. Seed 966 (161x97): camera 3000 units away; the reference's float
    evaluation "hits" a y-clipped cylinder 300 units beyond the end of its true piece -- the bound of a quadric with an open clip box now
    only holds near the quadric (rt_pack.h). Seed 9038 (97x161): a floor at t = 2992, then a quadric on its degenerate branch, whose inverted
    comparison (trap T4) accepts t = 22 925 > tmin and so moves the closest hit AWAY, then a cylinder at 3018 that the reference therefore
    shows -- a slab-table mask may not stop at the closest hit so far when a degenerate quadric is among the candidates (rt_device.h)."""
    W, H = {966: (161, 97), 9038: (97, 161)}.get(seed, [(96, 54), (97, 55)][seed % 2])
    sc = random_scenes.pencil_scene(seed, W, H)
    ref, cnt = oracle.OracleScene(sc, W, H, small_textures["textures"], small_textures["cubemap"], texture_lod=0).render()
    for cull in (1, 2, 0):
        img, hc = harness.render(sc, W, H, small_textures["textures"], small_textures["cubemap"], cull=cull)
        same = (img.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(img) & np.isnan(ref))
        assert same.all(), (seed, cull, int((~same).sum()))
        assert hc["closest"] == cnt["rays_closest"] and hc["shadow_ref"] == cnt["rays_shadow"], (seed, cull)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(24))
def test_scaled_quaternion_scene_on_gpu(built, small_textures, seed):
    from raytracing_opengl_amd import wrapper
    w, h = [(160, 96), (161, 97)][seed % 2]
    sc = random_scenes.scaled_quat_scene(seed, w, h)
    ref, cnt = oracle.OracleScene(sc, w, h, small_textures["textures"], small_textures["cubemap"], texture_lod=1).render()
    gl = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"])
    gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
    gl.draw()
    img = gl.read_pixels()
    st = gl.stats()
    gl.stop()
    v = parity_bar.judge(img, ref)      # absolute 1e-4 where |oracle| <= 1; non-unit quaternions scale normals: pixels can reach 1e13 -> relative above 1
    assert v["ok"], (seed, v)
    assert st["rays_closest"] == cnt["rays_closest"] and st["rays_shadow"] == cnt["rays_shadow"], seed


SWEEP = 150   # seeds per generator in the -m gpu suite (0.18 s each, nearly all of it the oracle on the box's host: round 3's 500 were 277 s of the
              # suite's 495, against a 1 200 s budget for everything the driver runs; tools/fuzz_gpu.py runs the 10^4-scale sweeps)


@pytest.mark.parametrize("seed", range(16))
def test_sized_torus_scene_bit_exact_on_host(built, small_textures, seed):
    """Tori of every size (tests/random_scenes.py::sized_torus_scene: R 0.04 .. 25, tubes of 4 % .. 140 % of it -- round 6): most of them lie
    outside the size range the cull premises were audited on and are never culled (rt_pack.h); those inside it take the noise-aware
    inflation. Culls on == culls off == oracle, bit for bit, ray counts included."""
    W, H = [(112, 64), (97, 65)][seed % 2]
    sc = random_scenes.sized_torus_scene(seed, W, H)
    ref, cnt = oracle.OracleScene(sc, W, H, small_textures["textures"], small_textures["cubemap"], texture_lod=0).render()
    for cull in (True, False):
        img, hc = harness.render(sc, W, H, small_textures["textures"], small_textures["cubemap"], cull=cull)
        same = (img.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(img) & np.isnan(ref))
        assert same.all(), (seed, cull, int((~same).sum()))
        assert hc["closest"] == cnt["rays_closest"] and hc["shadow_ref"] == cnt["rays_shadow"], (seed, cull)


@pytest.mark.gpu
@pytest.mark.parametrize("gen", ["random_scene", "nasty_scene", "scaled_quat_scene", "crowd_scene", "pencil_scene", "sized_torus_scene"])
def test_fuzz_sweep_on_gpu(built, small_textures, gen):
    """150 seeds of each generator (80 of the long-table ones) through ONE context per frame size (re-specialised per scene, as a program that swaps scenes
    would): culls on against the un-culled oracle -- max 1e-4, NaN/inf in the same places, identical ray counts (counting variant of the
    kernel: first-level culls); then the product variant (no counters; group culls and ray pencils where the scene has long tables)
    must reproduce the counting variant's frame bit for bit."""
    from raytracing_opengl_amd import wrapper
    make = getattr(random_scenes, gen)
    sizes = [(96, 64), (97, 65)]
    ctx = {}
    bad = []
    tally = dict(needed_relative=0, above_one=0, values=0)
    long_tables = gen in ("crowd_scene", "pencil_scene")
    for seed in range(20000, 20000 + (80 if long_tables else SWEEP)):   # long tables: ~30x the oracle time each
        w, h = sizes[seed % 2]
        sc = make(seed, w, h)
        ref, cnt = oracle.OracleScene(sc, w, h, small_textures["textures"], small_textures["cubemap"], texture_lod=1).render()
        gl = ctx.get((w, h))
        if gl is None:
            gl = ctx[(w, h)] = wrapper.make_renderer(sc, w, h, small_textures["textures"], small_textures["cubemap"])
        else:
            gl.init_shaders(sc.defines)
            gl.uploader = wrapper.SceneUploader(sc, gl)
            gl.uploader.init()
        gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 1)
        gl.draw()
        img = gl.read_pixels()
        st = gl.stats()
        gl.set_option(wrapper.RTX_OPT_COUNT_RAYS, 0)
        gl.draw()
        product = gl.read_pixels()
        v = parity_bar.judge(img, ref)   # absolute 1e-4 on colours up to 1; relative to the pixel where a degenerate scene's values exceed 1
        ok = v["ok"]
        for k in ("needed_relative", "above_one", "values"):
            tally[k] += v[k]
        ok = ok and st["rays_closest"] == cnt["rays_closest"] and st["rays_shadow"] == cnt["rays_shadow"]
        ok = ok and np.array_equal(product.view(np.uint32), img.view(np.uint32))
        if not ok:
            bad.append(seed)
    for gl in ctx.values():
        gl.stop()
    # how much of the verdict rests on the relative part of the bar (parity_bar.py): said, and kept with the run's output
    line = (f"{gen}: {tally['values']} channel values, {tally['above_one']} with |oracle| > 1, of which {tally['needed_relative']} differ by more than 1e-4 absolute "
            f"(inside 1e-4 relative); everything at or below 1 is within 1e-4 absolute")
    print(line)
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fuzz_bar_tally.txt"), "a") as f:
        f.write(line + "\n")
    assert not bad, f"{gen}: {len(bad)} scenes differ from the oracle or between kernel variants, seeds {bad[:20]}"
