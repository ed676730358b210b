"""Reference traps that no BASELINE scene reaches: hand-built scenes,
device logic (host build) against the oracle bit for bit, and the oracle's own event counters prove
each trap is really exercised."""
import numpy as np
import pytest

import harness
import trap_scenes
from oracle import oracle

EXPECT = {  # counters that must be non-zero in each scene
    "glass_tir": ("tir_breaks", "refract_segments", "box_inside_hits", "side_miss", "segment_cap_hits"),
    "inside_box": ("box_inside_hits", "box_nan_hits"),
    "degenerate_rings": ("t4_taken", "alpha_pass", "light_hits"),
    "planes_glass": ("refract_segments", "side_miss", "light_hits"),
}


@pytest.mark.parametrize("name", sorted(trap_scenes.ALL))
@pytest.mark.parametrize("cull", [True, False])
def test_trap_scene_bit_exact_on_host(built, small_textures, name, cull):
    w, h = (321, 181) if name == "inside_box" else (320, 180)  # odd size: some rays get exact-zero direction components
    sc = trap_scenes.ALL[name](w, h)
    ref, cnt = oracle.OracleScene(sc, w, h, small_textures["textures"], small_textures["cubemap"], texture_lod=0).render()
    for key in EXPECT[name]:
        assert cnt[key] > 0, f"{name}: trap counter {key} is zero -- the scene no longer exercises it"
    if name == "glass_tir":
        assert cnt["max_segments"] == 256  # the shared segment cap (refraction does i--, trap T2)
    img, hc = harness.render(sc, w, h, small_textures["textures"], small_textures["cubemap"], cull=cull)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert hc["closest"] == cnt["rays_closest"] and hc["shadow_ref"] == cnt["rays_shadow"]
