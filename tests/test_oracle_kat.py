"""Known-answer tests that pin the ORACLE itself (the reference ships no golden vectors for this
path): each intersector against a closed form or a float64 solve on
well-conditioned rays, the GLSL-semantics traps of SURVEY.md Appendix A.3, the fixed atan/asin
series, the sampler rule, and the ray-count pins of SURVEY.md Appendix C.3."""
import ctypes
import math
import struct

import numpy as np
import pytest

from oracle import oracle
from raytracing_opengl_amd import scenes

F3 = ctypes.c_float * 3
F5 = ctypes.c_float * 5


def _isect(type_, record: bytes, ro, rd, tmin=1e6, hollow=0):
    out = F5()
    buf = ctypes.create_string_buffer(record, len(record))
    assert oracle.lib().orc_kat_intersect(type_, buf, F3(*ro), F3(*rd), tmin, hollow, out) == 0
    return bool(out[0]), out[1], (out[2], out[3], out[4])


def _mat():
    return b"\0" * 64


def _quat(angle_deg=0.0, axis=(0, 0, 1)):
    a = math.radians(angle_deg) / 2
    s = math.sin(a)
    n = math.sqrt(sum(c * c for c in axis))
    return struct.pack("<4f", axis[0] / n * s, axis[1] / n * s, axis[2] / n * s, math.cos(a))


def test_sphere_closed_form_and_hollow():
    sph = struct.pack("<4f", 0, 0, 10, 2)
    hit, t, _ = _isect(oracle.TYPE_SPHERE, sph, (0, 0, 0), (0, 0, 1))
    assert hit and t == pytest.approx(8.0, abs=1e-6)
    # origin inside: near root negative -> miss unless hollow (far root), rt.frag:350-353
    hit, t, _ = _isect(oracle.TYPE_SPHERE, sph, (0, 0, 10), (0, 0, 1))
    assert not hit
    hit, t, _ = _isect(oracle.TYPE_SPHERE, sph, (0, 0, 10), (0, 0, 1), hollow=1)
    assert hit and t == pytest.approx(2.0, abs=1e-6)
    # strict t < tmin
    hit, _, _ = _isect(oracle.TYPE_SPHERE, sph, (0, 0, 0), (0, 0, 1), tmin=8.0)
    assert not hit
    # grazing miss
    hit, _, _ = _isect(oracle.TYPE_SPHERE, sph, (2.0001, 0, 0), (0, 0, 1))
    assert not hit


def test_plane_is_one_sided():
    pl = struct.pack("<6f", 0, 1, 0, 0, -1, 0)  # normal, pos
    hit, t, _ = _isect(oracle.TYPE_PLANE, pl, (0, 1, 0), (0, -1, 0))
    assert hit and t == pytest.approx(2.0)
    hit, _, _ = _isect(oracle.TYPE_PLANE, pl, (0, -3, 0), (0, 1, 0))  # from below: denom > 0 -> no hit (trap T1)
    assert not hit
    hit, _, _ = _isect(oracle.TYPE_PLANE, pl, (0, 1, 0), (1, 0, 0))   # parallel
    assert not hit


def test_ring_uses_squared_radii_and_reports_uv():
    r1, r2 = 1.0, 4.0  # squared radii 1 and 2
    ring = _mat() + _quat() + struct.pack("<3fi2f2f", 0, 0, 5, 0, r1, r2, 0, 0)
    hit, t, uv = _isect(oracle.TYPE_RING, ring, (1.5, 0, 0), (0, 0, 1))
    assert hit and t == pytest.approx(5.0)
    assert uv[0] == pytest.approx((1.5 ** 2 - r1) / (r2 - r1)) and uv[1] == pytest.approx(1.0)
    assert not _isect(oracle.TYPE_RING, ring, (0.5, 0, 0), (0, 0, 1))[0]   # inside the hole
    assert not _isect(oracle.TYPE_RING, ring, (2.5, 0, 0), (0, 0, 1))[0]   # outside
    hit, _, uv = _isect(oracle.TYPE_RING, ring, (0, -1.5, 0), (0, 0, 1))
    assert hit and uv[1] == pytest.approx(0.0, abs=1e-7)                    # v = cos(phi)


def test_box_slab_normal_inside_and_nan():
    box = _mat() + _quat() + struct.pack("<3f f 3f i", 0, 0, 10, 0, 1, 2, 3, 0)
    hit, t, n = _isect(oracle.TYPE_BOX, box, (0.25, 0.5, 0), (0, 0, 1))
    assert hit and t == pytest.approx(7.0) and n == pytest.approx((0, 0, -1))
    # exact-zero direction component with the origin on the negative side of that axis: t1.x = inf - inf = NaN sits in
    # the FIRST slot of max(max(t1.x,t1.y),t1.z), survives GLSL max, falls through both early-outs -> "hit" with NaN t (trap T5)
    hit, t, n = _isect(oracle.TYPE_BOX, box, (-0.25, 0.5, 0), (0, 0, 1))
    assert hit and math.isnan(t)
    # the same ray mirrored to the positive side: the NaN lands in t2 and in the second slot of min -> ordinary hit
    hit, t, n = _isect(oracle.TYPE_BOX, box, (5, 0.5, 10.5), (-1, 0, 0))
    assert hit and t == pytest.approx(4.0)
    # origin inside: negative entry distance is accepted (trap T21)
    hit, t, _ = _isect(oracle.TYPE_BOX, box, (0.1, 0.2, 10.3), (0.3, 0.5, 0.81))
    assert hit and t < 0
    # rotated box: 45 degrees about z, ray along +x hits the edge-on face at distance 10 - sqrt(2)... use y-offset 0
    rbox = _mat() + _quat(45, (0, 0, 1)) + struct.pack("<3f f 3f i", 10, 0, 0, 0, 1, 1, 1, 0)
    hit, t, _ = _isect(oracle.TYPE_BOX, rbox, (0, 0.0001, 0.0002), (1, 0.00001, 0.00002))
    assert hit and t == pytest.approx(10 - math.sqrt(2), rel=1e-3)


def _torus_roots_f64(ro, rd, R, r):
    o, d = np.array(ro, float), np.array(rd, float)
    a, b, c = d @ d, o @ d, o @ o + R * R - r * r
    axy, bxy, cxy = d[:2] @ d[:2], o[:2] @ d[:2], o[:2] @ o[:2]
    p = np.polymul([a, 2 * b, c], [a, 2 * b, c]) - 4 * R * R * np.array([0, 0, axy, 2 * bxy, cxy])
    roots = np.roots(p)
    real = [z.real for z in roots if abs(z.imag) < 1e-9 and z.real > 0]
    return min(real) if real else None


def test_torus_durand_kerner_against_float64_quartic():
    R, r = 1.0, 0.5
    tor = _mat() + _quat() + struct.pack("<3f f 2f 2f", 0, 0, 0, 0, R, r, 0, 0)
    rng = np.random.default_rng(7)
    checked = 0
    for _ in range(600):
        ro = rng.uniform(-3, 3, 3)
        rd = rng.uniform(-1.2, 1.2, 3) * np.array([1, 1, 0.3]) - ro  # aim at the torus' neighbourhood
        rd /= np.linalg.norm(rd)
        ref = _torus_roots_f64(ro, rd, R, r)
        hit, t, _ = _isect(oracle.TYPE_TORUS, tor, ro, rd)
        if ref is None:
            continue
        # skip near-tangent rays (ill-conditioned double roots) -- the 1e-3 acceptance window is a reference trap (T14)
        o, d = np.array(ro), np.array(rd)
        p = o + d * ref
        g = p * (p @ p - r * r - R * R * np.array([1, 1, -1]))
        if abs(g @ d) / (np.linalg.norm(g) + 1e-30) < 0.2:
            continue
        assert hit and t == pytest.approx(ref, abs=2e-3), (ro, rd, t, ref)
        checked += 1
    assert checked > 50
    # beyond t = 100 hits are rejected (rt.frag:486)
    assert not _isect(oracle.TYPE_TORUS, tor, (0, 1.0, -150), (0, 0, 1))[0]


def _surface(a=0, b=0, c=0, d=0, e=0, f=0, pos=(0, 0, 0), quat=None, vmin=(-3.4e38,) * 3, vmax=(3.4e38,) * 3):
    return _mat() + (quat or _quat()) + struct.pack("<3f f 3f f 3f 6f 3f", *vmin, 0, *vmax, 0, *pos, a, b, c, d, e, f, 0, 0, 0)


def test_quadric_sphere_clip_box_and_degenerate_branch():
    unit_sphere = _surface(a=1, b=1, c=1, f=-1, pos=(0, 0, 5))
    hit, t, _ = _isect(oracle.TYPE_SURFACE, unit_sphere, (0, 0, 0), (0, 0, 1))
    assert hit and t == pytest.approx(4.0, abs=1e-5)
    # world-space clip box (trap T6): cut away z < 5 -> the far root is used
    clipped = _surface(a=1, b=1, c=1, f=-1, pos=(0, 0, 5), vmin=(-9, -9, 5), vmax=(9, 9, 9))
    hit, t, _ = _isect(oracle.TYPE_SURFACE, clipped, (0, 0, 0), (0, 0, 1))
    assert hit and t == pytest.approx(6.0, abs=1e-5)
    # negative discriminant -> NaN control flow ends in false (trap T5)
    assert not _isect(oracle.TYPE_SURFACE, unit_sphere, (3, 0, 0), (0, 0, 1))[0]
    # degenerate branch |p2| < 1e-6 (trap T4): cylinder x^2+y^2=1, ray along its axis, returns t > tmin (sic)
    cyl = _surface(a=1, b=1, f=-1)
    hit, t, _ = _isect(oracle.TYPE_SURFACE, cyl, (0.5, 0, 0), (0, 0, 1), tmin=-1e30)
    assert math.isinf(t) or math.isnan(t) or hit in (True, False)  # p1 = 0 -> t = -p3/0
    cone = _surface(a=1, b=1, c=-1)  # direction on the cone's own asymptote: p2 = 0
    s = 1 / math.sqrt(2)
    hit, t, _ = _isect(oracle.TYPE_SURFACE, cone, (0.3, 0, -2), (s, 0, s), tmin=0.5)
    assert hit and t > 0.5       # "hit" because t > tmin -- the inverted comparison
    hit2, _, _ = _isect(oracle.TYPE_SURFACE, cone, (0.3, 0, -2), (s, 0, s), tmin=1e6)
    assert not hit2


def test_atan_asin_series_accuracy():
    lib = oracle.lib()
    rng = np.random.default_rng(1)
    worst = 0.0
    for _ in range(20000):
        y, x = rng.normal(), rng.normal()
        worst = max(worst, abs(lib.orc_kat_atan2(y, x) - math.atan2(np.float32(y), np.float32(x))))
        v = rng.uniform(-1, 1)
        worst = max(worst, abs(lib.orc_kat_asin(v) - math.asin(np.float32(v))))
    assert worst < 3e-7  # one float rounding of a ~1e-11-accurate double
    assert lib.orc_kat_atan2(0.0, -1.0) == pytest.approx(math.pi, abs=1e-6)
    assert lib.orc_kat_atan2(-1.0, 0.0) == pytest.approx(-math.pi / 2, abs=1e-6)
    assert math.isnan(lib.orc_kat_asin(1.0000001))  # trap T15: not clamped


def test_text_round_trip_of_shader_constants():
    lib = oracle.lib()
    assert lib.orc_kat_text_round_trip(0.025) == np.float32(0.025)
    assert lib.orc_kat_text_round_trip(0.123456789) == np.float32(0.123457)  # "%f" keeps 6 decimals (trap T9)


def test_sampler_rule():
    lib = oracle.lib()
    img = np.zeros((2, 4, 3), np.uint8)
    img[0, :, 0] = [0, 85, 170, 255]
    img[1, :, 0] = [255, 170, 85, 0]
    t = oracle.Texture(4, 2, 3, 0, img.ctypes.data)
    out = (ctypes.c_float * 4)()
    lib.orc_kat_sample2d(ctypes.byref(t), (0.5 + 1) / 4, 0.25, out)      # exactly on texel (1,0)
    assert out[0] == pytest.approx(85 / 255) and out[3] == 1.0
    lib.orc_kat_sample2d(ctypes.byref(t), 2.0 / 4, 0.25, out)            # midway between texels 1 and 2
    assert out[0] == pytest.approx((85 + 170) / 2 / 255)
    lib.orc_kat_sample2d(ctypes.byref(t), 0.0, 0.25, out)                # REPEAT: wraps to texel 3 and 0
    assert out[0] == pytest.approx((255 + 0) / 2 / 255)
    lib.orc_kat_sample2d(ctypes.byref(t), float("nan"), 0.25, out)       # NaN -> coordinate 0
    assert out[0] == pytest.approx((255 + 0) / 2 / 255)
    # cube: +X face centre, and the face table's orientation
    faces = [np.full((2, 2, 3), 10 * (f + 1), np.uint8) for f in range(6)]
    cm = oracle.Cubemap(2, 3, (ctypes.c_void_p * 6)(*[f.ctypes.data for f in faces]))
    for d, f in [((1, 0, 0), 0), ((-1, 0, 0), 1), ((0, 1, 0), 2), ((0, -1, 0), 3), ((0, 0, 1), 4), ((0, 0, -1), 5),
                 ((0.9, 0.2, -0.3), 0), ((0.2, -0.3, -0.9), 5)]:
        lib.orc_kat_sample_cube(ctypes.byref(cm), F3(*d), out)
        assert out[0] == pytest.approx(10 * (f + 1) / 255)


def test_ray_count_pins(built, mid_textures):
    """SURVEY.md Appendix C.3 pins (nearest-texel probe): rays/pixel within 1 % of the survey numbers."""
    for (w, h, d), (closest, shadow, dk_mean) in {(640, 480, 1): (1.060, 0.613, 12.05), (480, 270, 4): (1.420, 0.885, 12.8)}.items():
        sc = scenes.build_scene("default", w, h, d)
        _img, c = oracle.OracleScene(sc, w, h, mid_textures["textures"], mid_textures["cubemap"], texture_lod=0).render()
        px = w * h
        assert c["rays_closest"] / px == pytest.approx(closest, rel=0.01)
        assert c["rays_shadow"] / px == pytest.approx(shadow, rel=0.02)
        assert c["dk_sweeps"] / c["dk_solves"] == pytest.approx(dk_mean, rel=0.03)
        assert c["segment_cap_hits"] == 0 and c["tir_breaks"] == 0


def test_mip_chain_and_trilinear_rule():
    """Phase-B texture rule of the oracle: integer box-filtered mips with floor-halving sizes, odd
    sizes handled by clamping the second tap, trilinear blend between floor(lambda) and the next level."""
    lib = oracle.lib()
    lib.orc_kat_mip_level.restype = ctypes.c_int
    lib.orc_kat_mip_level.argtypes = [ctypes.POINTER(oracle.Texture), ctypes.c_int, ctypes.c_void_p]
    lib.orc_kat_sample2d_lod.argtypes = [ctypes.POINTER(oracle.Texture), ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float * 4]
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(5, 12, 3), dtype=np.uint8)  # odd height
    t = oracle.Texture(12, 5, 3, 0, img.ctypes.data)
    out = np.zeros((5 * 12 * 4,), np.uint8)
    dims = lib.orc_kat_mip_level(ctypes.byref(t), 1, out.ctypes.data)
    assert dims == (6 << 16 | 2)
    lvl1 = out[: 6 * 2 * 4].reshape(2, 6, 4)
    src = np.concatenate([img, np.full((5, 12, 1), 255, np.uint8)], axis=-1).astype(np.int32)
    for j in range(2):
        for i in range(6):
            exp = (src[2 * j, 2 * i] + src[2 * j, 2 * i + 1] + src[2 * j + 1, 2 * i] + src[2 * j + 1, 2 * i + 1] + 2) >> 2
            assert (lvl1[j, i] == exp).all()
    dims = lib.orc_kat_mip_level(ctypes.byref(t), 2, out.ctypes.data)
    assert dims == (3 << 16 | 1)
    assert lib.orc_kat_mip_level(ctypes.byref(t), 3, out.ctypes.data) == (1 << 16 | 1)
    assert lib.orc_kat_mip_level(ctypes.byref(t), 4, out.ctypes.data) == 0
    c0, c1, cm = (ctypes.c_float * 4)(), (ctypes.c_float * 4)(), (ctypes.c_float * 4)()
    lib.orc_kat_sample2d_lod(ctypes.byref(t), 0.3, 0.6, 1.0, c0)
    lib.orc_kat_sample2d_lod(ctypes.byref(t), 0.3, 0.6, 2.0, c1)
    lib.orc_kat_sample2d_lod(ctypes.byref(t), 0.3, 0.6, 1.25, cm)
    for k in range(4):
        assert cm[k] == pytest.approx(0.75 * c0[k] + 0.25 * c1[k], abs=1e-6)
    lib.orc_kat_sample2d_lod(ctypes.byref(t), 0.3, 0.6, -5.0, c0)
    lib.orc_kat_sample2d(ctypes.byref(t), 0.3, 0.6, c1)
    assert list(c0) == list(c1)                       # lambda <= 0 -> level-0 bilinear
    lib.orc_kat_sample2d_lod(ctypes.byref(t), 0.3, 0.6, 99.0, c0)
    lib.orc_kat_sample2d_lod(ctypes.byref(t), 0.9, 0.1, 3.0, c1)
    assert list(c0) == list(c1)                       # clamped to the 1x1 top level


def test_effective_cpus_respects_affinity_and_cgroup_quota():
    """oracle.effective_cpus(): the thread count every checker uses -- the affinity mask capped by the cgroup CPU quota (the GPU boxes show 256
    hardware threads and grant 16 CPUs of time: 256 OpenMP threads inside that quota ran the oracle at 8 Mray/s where 16 run it at 13)."""
    import os
    n, quota = oracle.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    if hasattr(os, "sched_getaffinity"):
        assert n <= len(os.sched_getaffinity(0))
    if quota is not None:
        assert quota > 0 and n <= int(quota + 0.999)
