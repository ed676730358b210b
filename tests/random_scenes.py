"""Seeded random scenes for fuzz-style parity tests: every primitive class, random rotations (some identity), random
materials (diffuse / reflective / refractive / alpha-textured rings), 0-2 lights of each kind, camera placed outside or
inside the cluster. Built only from tests/scene_util.py records, i.e. the same std140 bytes the reference's host writes."""
import math

import numpy as np

from scene_util import (box, light_direct, light_point, make_scene, material, plane, quat_euler, ring, sphere, surface, torus)


def _mat(rng):
    kind = rng.random()
    color = tuple(rng.random(3) * 0.9 + 0.1)
    spec = int(rng.integers(0, 200)) if rng.random() < 0.8 else 0
    if kind < 0.5:
        return material(color, spec, 0.0)
    if kind < 0.85:
        return material(color, spec, float(rng.random() * 0.6 + 0.05))
    return material(color, spec, float(rng.random() * 0.3), float(rng.choice([0.7, 1.0, 1.125, 1.33, 1.5, 1.9])), tuple(rng.random(3) * 0.5), 1.0)


def _quat(rng):
    return (0.0, 0.0, 0.0, 1.0) if rng.random() < 0.35 else quat_euler(*(rng.random(3) * 2 * math.pi - math.pi))


def _pos(rng, spread=4.0, z0=6.0):
    return (float(rng.normal() * spread * 0.6), float(rng.normal() * spread * 0.35), float(z0 + rng.normal() * spread * 0.5))


def random_scene(seed: int, w: int, h: int):
    rng = np.random.default_rng(seed)
    depth = int(rng.integers(1, 6))
    spheres = [sphere(_pos(rng), float(rng.random() * 0.9 + 0.2), _mat(rng), hollow=bool(rng.random() < 0.3), quat=_quat(rng),
                      texture=int(rng.choice([0, 0, 1, 2, 3]))) for _ in range(int(rng.integers(0, 5)))]
    planes = [plane((0, 1, 0), (0, -2.0 - float(rng.random()), 0), _mat(rng))] if rng.random() < 0.5 else []
    if rng.random() < 0.2:
        planes.append(plane(tuple(rng.normal(size=3)), _pos(rng, 6.0, 10.0), _mat(rng)))
    boxes = [box(_pos(rng), tuple(rng.random(3) * 1.2 + 0.2), _mat(rng), quat=_quat(rng), texture=int(rng.choice([0, 0, 5])))
             for _ in range(int(rng.integers(0, 4)))]
    toruses = [torus(_pos(rng), float(rng.random() * 0.8 + 0.5), float(rng.random() * 0.3 + 0.1), _mat(rng), quat=_quat(rng))
               for _ in range(int(rng.integers(0, 3)))]
    rings = [ring(_pos(rng), float(rng.random() * 0.5 + 0.2), float(rng.random() * 1.0 + 0.9), _mat(rng) if rng.random() < 0.5 else material((0, 0, 0), 0, 0),
                  quat=_quat(rng), texture=int(rng.choice([0, 4]))) for _ in range(int(rng.integers(0, 3)))]
    surfaces = []
    for _ in range(int(rng.integers(0, 4))):
        p = _pos(rng)
        kind = int(rng.integers(0, 4))
        clip = dict(vmin=(p[0] - 1.5, p[1] - 1.5, p[2] - 1.5), vmax=(p[0] + 1.5, p[1] + 1.5, p[2] + 1.5)) if rng.random() < 0.7 else {}
        coef = [dict(a=1, b=1, c=1, f=-float(rng.random() + 0.3)),            # ellipsoid
                dict(a=1, b=-1, c=1),                                         # cone
                dict(a=float(rng.random() + 0.5), b=float(rng.random() + 0.5), f=-0.5),   # elliptic cylinder
                dict(a=0.5, b=-0.5, d=-1)][kind]                              # saddle
        surfaces.append(surface(p, _mat(rng), quat=_quat(rng), **coef, **clip))
    lights_point = [light_point(_pos(rng, 3.0, 3.0), float(rng.random() * 0.3 + 0.05), intensity=float(rng.random() * 30 + 5)) for _ in range(int(rng.integers(0, 3)))]
    lights_direct = [light_direct(tuple(rng.normal(size=3) + np.array([0, -1.5, 0]))) for _ in range(int(rng.integers(0, 3)))]
    inside = rng.random() < 0.15 and boxes
    cam = _pos(rng, 1.0, -3.0) if not inside else (0.0, 0.0, 6.0)
    cam_quat = quat_euler(float(rng.normal() * 0.15), float(rng.normal() * 0.3), 0.0) if rng.random() < 0.6 else (0.0, 0.0, 0.0, 1.0)
    return make_scene(w, h, depth, spheres=spheres, planes=planes, surfaces=surfaces, boxes=boxes, toruses=toruses, rings=rings,
                      lights_point=lights_point, lights_direct=lights_direct, cam_pos=cam, cam_quat=cam_quat)


def nasty_scene(seed: int, w: int, h: int):
    """Degenerate and extreme configurations on purpose: coincident and nested primitives (exact ties), zero and huge radii,
    far 'planets', self-intersecting tori (r > R), rings with swapped radii, flat boxes, quadrics with vanishing
    coefficients, refraction index exactly 1, lights inside objects, axis-aligned cameras on integer coordinates."""
    rng = np.random.default_rng(seed ^ 0x5eed)
    depth = int(rng.integers(1, 9))
    grid = lambda: tuple(float(v) for v in rng.integers(-3, 4, 3) + np.array([0, 0, 6]))   # integer lattice positions: exact ties
    def mat():
        k = rng.random()
        color = tuple(rng.choice([0.0, 0.5, 1.0], 3))
        spec = int(rng.choice([0, 1, 50, 1000]))
        if k < 0.4: return material(color, spec, 0.0)
        if k < 0.7: return material(color, spec, float(rng.choice([0.05, 0.5, 1.0])))
        return material(color, spec, float(rng.choice([0.0, 0.2, 1.0])), float(rng.choice([1.0, 1.0001, 0.5, 1.5, 3.0])), tuple(rng.choice([0.0, 0.3, 5.0], 3)), 1.0)
    q_choices = [(0.0, 0.0, 0.0, 1.0), quat_euler(math.pi / 2, 0, 0), quat_euler(0, math.pi / 2, 0), quat_euler(0, 0, math.pi), quat_euler(0.3, 0.2, 0.1)]
    quat = lambda: q_choices[int(rng.integers(len(q_choices)))]
    spheres = [sphere(grid(), float(rng.choice([0.0, 1e-3, 0.5, 1.0, 2.5])), mat(), hollow=bool(rng.random() < 0.4), quat=quat(), texture=int(rng.choice([0, 1, 7])))
               for _ in range(int(rng.integers(0, 5)))]
    if rng.random() < 0.3:
        spheres.append(sphere((float(rng.normal() * 2e4), float(rng.normal() * 5e3), 3.0e4), float(rng.choice([500.0, 5000.0])), mat(), texture=int(rng.choice([0, 2]))))
    if spheres and rng.random() < 0.4:
        spheres.append(spheres[0])                       # exact duplicate: equal t, first wins
    planes = []
    if rng.random() < 0.6: planes.append(plane((0, 1, 0), (0, float(rng.choice([-1.0, -2.0, 0.0])), 0), mat()))
    if rng.random() < 0.2: planes.append(plane((0, 0, -1), (0, 0, 9.0), mat()))
    boxes = [box(grid(), tuple(float(v) for v in rng.choice([0.0, 0.5, 1.0, 4.0], 3)), mat(), quat=quat(), texture=int(rng.choice([0, 5]))) for _ in range(int(rng.integers(0, 4)))]
    toruses = [torus(grid(), float(rng.choice([0.3, 1.0, 1.0, 2.0])), float(rng.choice([0.0, 0.25, 1.0, 1.5])), mat(), quat=quat()) for _ in range(int(rng.integers(0, 3)))]
    rings = [ring(grid(), float(rng.choice([0.0, 0.5, 1.5])), float(rng.choice([0.4, 1.0, 2.0])), mat(), quat=quat(), texture=int(rng.choice([0, 4]))) for _ in range(int(rng.integers(0, 3)))]
    surfaces = []
    for _ in range(int(rng.integers(0, 4))):
        p = grid()
        coef = dict(a=float(rng.choice([0.0, 1.0, -1.0, 0.02])), b=float(rng.choice([0.0, 1.0, -1.0])), c=float(rng.choice([0.0, 1.0, -1.0])),
                    d=float(rng.choice([0.0, -1.0])), e=float(rng.choice([0.0, 0.5])), f=float(rng.choice([0.0, -1.0, 1.0])))
        clip = dict(vmin=(p[0] - 2, p[1] - 2, p[2] - 2), vmax=(p[0] + 2, p[1] + 2, p[2] + 2)) if rng.random() < 0.6 else {}
        surfaces.append(surface(p, mat(), quat=quat(), **coef, **clip))
    lights_point = [light_point(grid(), float(rng.choice([0.0, 0.1, 1.0])), intensity=float(rng.choice([0.0, 10.0, 40.0]))) for _ in range(int(rng.integers(0, 3)))]
    lights_direct = [light_direct(tuple(float(v) for v in rng.choice([0.0, 1.0, -1.0], 3) + np.array([0, -1e-3, 0]))) for _ in range(int(rng.integers(0, 3)))]
    cam = tuple(float(v) for v in rng.integers(-2, 3, 3) + np.array([0, 0, -3])) if rng.random() < 0.7 else grid()
    cam_quat = (0.0, 0.0, 0.0, 1.0) if rng.random() < 0.6 else quat_euler(0.0, float(rng.choice([0.0, math.pi / 2, math.pi, 0.3])), 0.0)
    return make_scene(w, h, depth, spheres=spheres, planes=planes, surfaces=surfaces, boxes=boxes, toruses=toruses, rings=rings,
                      lights_point=lights_point, lights_direct=lights_direct, cam_pos=cam, cam_quat=cam_quat)


# byte offset of quat_rotation inside each record (include/rtx/scene.h) and the record size
_QUAT_AT = {"spheres_buf": (80, 112), "surfaces_buf": (64, 160), "boxes_buf": (64, 112), "toruses_buf": (64, 112), "rings_buf": (64, 112)}


def scaled_quat_scene(seed: int, w: int, h: int):
    """random_scene(seed) with about half of its rotation quaternions (and sometimes the camera's) scaled to a norm other
    than 1. rt.frag's rotate() is q v conj(q) (rt.frag:306-311), so such a quaternion also scales the primitive's local
    frame by |q|^2: spheres keep their shape (only their texture lookup rotates), boxes / tori / rings / quadrics change
    size in world space and see non-unit ray directions. Nothing in the reference normalises the field, so the result is
    defined and has to be reproduced (advisor finding, round 1: cull bounds had assumed unit quaternions)."""
    import struct
    sc = random_scene(seed, w, h)
    rng = np.random.default_rng(seed ^ 0x9a7)
    blocks = dict(sc.blocks)
    for name, (off, size) in _QUAT_AT.items():
        buf = bytearray(blocks[name])
        for i in range(len(buf) // size):
            if rng.random() < 0.5:
                k = float(rng.choice([0.6, 0.8, 0.9, 0.97, 0.9995, 1.0005, 1.05, 1.2, 1.5]))
                q = [np.float32(v) * np.float32(k) for v in struct.unpack_from("<4f", buf, i * size + off)]
                struct.pack_into("<4f", buf, i * size + off, *q)
        blocks[name] = bytes(buf)
    if rng.random() < 0.25:
        buf = bytearray(blocks["scene_buf"])
        q = [np.float32(v) * np.float32(rng.choice([0.8, 1.25])) for v in struct.unpack_from("<4f", buf, 0)]
        struct.pack_into("<4f", buf, 0, *q)
        blocks["scene_buf"] = bytes(buf)
    return type(sc)(defines=sc.defines, blocks=blocks)


def crowd_scene(seed: int, w: int, h: int, lights=None, camera=None, planes_override=None):
    """Long primitive tables (16 ... 100 quadrics and / or tori, plus a few of everything else): exercises the second-level group culls
    (rt_device.h RT_GROUP) -- groups with an unbounded member, members with non-unit quaternions, quadrics of every kind incl. ones whose
    degenerate branch (trap T4) fires for axis-parallel rays, clusters spread far apart so that whole groups are skipped."""
    rng = np.random.default_rng(seed ^ 0xc0de)
    depth = int(rng.integers(1, 5))
    n_surf = int(rng.choice([0, 17, 40, 96]))
    n_tor = int(rng.choice([0, 16, 33, 64])) if n_surf else int(rng.choice([16, 33, 64]))

    def where(k, n):
        cols = max(1, int(math.sqrt(n) * 1.4))
        return (float((k % cols - cols / 2) * 2.7 + rng.normal() * 0.3), float((k // cols - n / cols / 2) * 2.9 + rng.normal() * 0.3), float(14.0 + rng.random() * 8.0))
    surfaces = []
    for k in range(n_surf):
        p = where(k, n_surf)
        coef = [dict(a=1, b=1, c=1, f=-float(rng.random() * 0.5 + 0.3)), dict(a=4, b=4, c=-1), dict(a=4, b=4, f=-1), dict(a=1.5, b=1.5, d=-1),
                dict(a=1.5, b=-1.5, d=-1), dict(a=4, b=4, c=-1, f=-1)][int(rng.integers(6))]
        r = rng.random()
        if r < 0.8:
            clip = dict(vmin=(p[0] - 1.2, p[1] - 1.2, p[2] - 1.2), vmax=(p[0] + 1.2, p[1] + 1.2, p[2] + 1.2))
        elif r < 0.9:
            from scene_util import FLT_MAX
            clip = dict(vmin=(-FLT_MAX, p[1] - 1.0, -FLT_MAX), vmax=(FLT_MAX, p[1] + 1.0, FLT_MAX))    # clipped in y only
        else:
            clip = {}                                                                                      # unbounded: its group is never culled
        q = _quat(rng)
        if rng.random() < 0.1:
            q = tuple(float(np.float32(v) * np.float32(0.9)) for v in q)
        surfaces.append(surface(p, _mat(rng), quat=q, **coef, **clip))
    toruses = []
    for k in range(n_tor):
        p = where(k, n_tor)
        q = _quat(rng)
        if rng.random() < 0.1:
            q = tuple(float(np.float32(v) * np.float32(1.1)) for v in q)
        toruses.append(torus((p[0], p[1], p[2] - 4.0), float(rng.random() * 0.5 + 0.6), float(rng.choice([0.0, 0.2, 0.3])), _mat(rng), quat=q))
    spheres = [sphere(_pos(rng, 8.0, 12.0), float(rng.random() + 0.3), _mat(rng)) for _ in range(int(rng.integers(0, 4)))]
    boxes = [box(_pos(rng, 8.0, 12.0), tuple(rng.random(3) + 0.3), _mat(rng), quat=_quat(rng)) for _ in range(int(rng.integers(0, 3)))]
    planes = [plane((0, 1, 0), (0, -12.0, 0), _mat(rng))] if rng.random() < 0.5 else []
    lights_point = [light_point((3.0, 5.0, 0.0), 0.1, intensity=25.5)]
    lights_direct = [light_direct((3.0, -1.0, 1.0))] if rng.random() < 0.7 else []
    cam_quat = quat_euler(float(rng.normal() * 0.1), float(rng.normal() * 0.2), 0.0) if rng.random() < 0.5 else (0.0, 0.0, 0.0, 1.0)
    cam_pos = (0.0, 0.0, -5.0)
    if lights is not None:      # pencil_scene: its own lights and camera, drawn after everything else so that the crowd stays the same
        lights_point, lights_direct = lights
    if camera is not None:
        cam_pos, cam_quat = camera
    if planes_override is not None:
        planes = planes_override
    return make_scene(w, h, depth, spheres=spheres, planes=planes, surfaces=surfaces, boxes=boxes, toruses=toruses, lights_point=lights_point,
                      lights_direct=lights_direct, cam_pos=cam_pos, cam_quat=cam_quat)


def pencil_scene(seed: int, w: int, h: int):
    """crowd_scene's long tables under lights and cameras chosen to stress the ray pencils (rt_scene_dev.h DevPencil): 0-4 point lights --
    in front of, inside and behind the crowd, inside a primitive's bound, thousands of units away (no pencil: precision), more lights
    than pencils -- 0-3 directional lights incl. axis-parallel ones and the zero vector, cameras inside the crowd, far away, rotated."""
    rng = np.random.default_rng(seed ^ 0x9e2c11)
    spots = [(3.0, 5.0, 0.0), (0.0, 0.0, 15.0), (0.5, -0.3, 18.0), (-6.0, 9.0, 30.0), (2500.0, 900.0, -4000.0), (0.0, 40.0, 14.0), (1.0e5, 0.0, 0.0)]
    n_point = int(rng.choice([0, 1, 1, 2, 4, 9]))
    lights_point = []
    for _ in range(n_point):
        p = spots[int(rng.integers(len(spots)))]
        p = tuple(float(v + rng.normal() * 0.5) for v in p)
        lights_point.append(light_point(p, float(rng.choice([0.1, 0.5])), intensity=float(rng.choice([25.5, 400.0]))))
    dirs = [(3.0, -1.0, 1.0), (0.0, -1.0, 0.0), (0.0, 0.0, 1.0), (1.0, 0.0, 0.0), (0.0, 0.0, 0.0), (-0.3, -0.2, -1.0)]
    lights_direct = [light_direct(dirs[int(rng.integers(len(dirs)))]) for _ in range(int(rng.choice([0, 1, 1, 2, 3])))]
    cams = [(0.0, 0.0, -5.0), (0.3, -0.4, 15.0), (0.0, 0.0, 40.0), (20.0, 3.0, 14.0), (0.0, 0.0, -3000.0)]
    cam_pos = cams[int(rng.integers(len(cams)))]
    yaw = {2: math.pi, 3: -math.pi / 2}.get(cams.index(cam_pos), 0.0)
    cam_quat = quat_euler(float(rng.normal() * 0.1), float(yaw + rng.normal() * 0.2), 0.0)
    # planes (drawn last: the scenes of earlier seeds keep their lights and cameras): in two scenes of three 1-3 mirrors / matt planes with
    # tilted, non-unit or zero normals -- floors, walls behind the crowd, a plane through the crowd, a plane behind the camera -- for the
    # mirror-image pencils of reflecting planes
    planes = None
    if rng.random() < 0.67:
        planes = []
        for _ in range(int(rng.integers(1, 4))):
            nrm = [(0.0, 1.0, 0.0), (0.0, 2.5, 0.0), (0.1, 1.0, -0.2), (0.0, 0.0, -1.0), (1.0, 0.3, 0.0), (0.0, -1.0, 0.0), (0.0, 0.0, 0.0), (0.6, 0.8, 0.0),
                   (0.0, 0.6, -0.8), (-0.36, 0.8, -0.48)][int(rng.integers(10))]
            pos = [(0.0, -12.0, 0.0), (0.0, -3.0, 16.0), (0.0, 0.0, 30.0), (-15.0, 0.0, 15.0), (0.0, 14.0, 0.0), (0.0, 0.0, -8.0)][int(rng.integers(6))]
            m = _mat(rng)
            if rng.random() < 0.6:      # a plain mirror (no refraction)
                m = material(tuple(rng.random(3) * 0.9 + 0.1), int(rng.integers(0, 200)), float(rng.choice([0.1, 0.5, 1.0])))
            planes.append(plane(nrm, pos, m))
    return crowd_scene(seed, w, h, lights=(lights_point, lights_direct), camera=(cam_pos, cam_quat), planes_override=planes)


def sized_torus_scene(seed: int, w: int, h: int):
    """Tori of every SIZE (round 6: the torus cull premises are statements about a float iteration whose behaviour depends on the torus' size -- rt_pack.h RT_TORUS_CULL_*): major radius
    log-uniform over 0.04 .. 25, tube 4 % .. 140 % of it (ring, horn and spindle tori), random rotations, a floor, one light of each kind. The other
    generators draw R in 0.3 .. 2 only. Used by tools/cull_audit.py (torus families) and the fuzz tests."""
    rng = np.random.default_rng(seed ^ 0x51ced)
    depth = int(rng.integers(1, 5))
    toruses = []
    for _ in range(int(rng.integers(3, 7))):
        R = float(np.exp(rng.uniform(np.log(0.04), np.log(25.0))))
        r = float(R * np.exp(rng.uniform(np.log(0.04), np.log(1.4))))
        p = (float(rng.normal() * (4.0 + 2.0 * R)), float(rng.normal() * (2.0 + R)), float(8.0 + 3.0 * R + rng.normal() * 3.0))
        toruses.append(torus(p, R, r, _mat(rng), quat=_quat(rng)))
    spheres = [sphere(_pos(rng, 6.0, 9.0), float(rng.random() * 0.9 + 0.2), _mat(rng)) for _ in range(int(rng.integers(0, 3)))]
    planes = [plane((0, 1, 0), (0, -30.0, 0), _mat(rng))] if rng.random() < 0.5 else []
    lights_point = [light_point((3.0, 15.0, -2.0), 0.1, intensity=60.0)]
    lights_direct = [light_direct((1.0, -2.0, 1.5))]
    return make_scene(w, h, depth, spheres=spheres, planes=planes, toruses=toruses, lights_point=lights_point, lights_direct=lights_direct,
                      cam_pos=(0.0, 0.0, -6.0), cam_quat=(0.0, 0.0, 0.0, 1.0))
