"""Hand-built scenes for edge-case tests: a tiny Python writer of the std140 records
(byte layouts: include/rtx/scene.h). Test infrastructure only."""
from __future__ import annotations

import math
import struct

from raytracing_opengl_amd.scenes import SceneBlocks

FLT_MAX = 3.402823466e38


def material(color=(1, 1, 1), specular=50, reflect=0.0, refract=0.0, absorb=(0, 0, 0), diffuse=0.7, kd=0.8, ks=0.2) -> bytes:
    return struct.pack("<3f f 3f f f f i f f 3f", *color, 0, *absorb, diffuse, reflect, refract, specular, kd, ks, 0, 0, 0)


def quat_euler(pitch=0.0, yaw=0.0, roll=0.0):
    """GLM quat(vec3 euler) in x,y,z,w memory order (float64 here; only used for hand-built test scenes)."""
    c = [math.cos(a / 2) for a in (pitch, yaw, roll)]
    s = [math.sin(a / 2) for a in (pitch, yaw, roll)]
    w = c[0] * c[1] * c[2] + s[0] * s[1] * s[2]
    x = s[0] * c[1] * c[2] - c[0] * s[1] * s[2]
    y = c[0] * s[1] * c[2] + s[0] * c[1] * s[2]
    z = c[0] * c[1] * s[2] - s[0] * s[1] * c[2]
    return (x, y, z, w)


IDENT = (0.0, 0.0, 0.0, 1.0)


def sphere(center, radius, mat, hollow=False, quat=IDENT, texture=0) -> bytes:
    return mat + struct.pack("<4f 4f i i 2f", *center, radius, *quat, texture, 1 if hollow else 0, 0, 0)


def plane(normal, pos, mat) -> bytes:
    return mat + struct.pack("<3f f 3f f", *pos, 0, *normal, 0)


def box(pos, form, mat, quat=IDENT, texture=0) -> bytes:
    return mat + struct.pack("<4f 3f f 3f i", *quat, *pos, 0, *form, texture)


def torus(pos, R, r, mat, quat=IDENT) -> bytes:
    return mat + struct.pack("<4f 3f f 2f 2f", *quat, *pos, 0, R, r, 0, 0)


def ring(pos, r_in, r_out, mat, quat=IDENT, texture=0) -> bytes:
    return mat + struct.pack("<4f 3f i 2f 2f", *quat, *pos, texture, r_in * r_in, r_out * r_out, 0, 0)


def surface(pos, mat, a=0, b=0, c=0, d=0, e=0, f=0, quat=IDENT, vmin=(-FLT_MAX,) * 3, vmax=(FLT_MAX,) * 3) -> bytes:
    return mat + struct.pack("<4f 3f f 3f f 3f 6f 3f", *quat, *vmin, 0, *vmax, 0, *pos, a, b, c, d, e, f, 0, 0, 0)


def light_point(pos, radius=0.1, color=(1, 1, 1), intensity=25.5, linear_k=0.22, quadratic_k=0.2) -> bytes:
    return struct.pack("<4f 3f f 2f 2f", *pos, radius, *color, intensity, linear_k, quadratic_k, 0, 0)


def light_direct(direction, color=(1, 1, 1), intensity=1.5) -> bytes:
    return struct.pack("<3f f 3f f", *direction, 0, *color, intensity)


def scene_block(width, height, cam_pos=(0, 0, -5), cam_quat=IDENT, depth=5) -> bytes:
    return struct.pack("<4f 3f f 3f i i i 2f", *cam_quat, *cam_pos, 0, 0, 0, 0, width, height, depth, 0, 0)


def make_scene(width, height, depth, spheres=(), planes=(), surfaces=(), boxes=(), toruses=(), rings=(), lights_point=(), lights_direct=(),
               cam_pos=(0, 0, -5), cam_quat=IDENT, ambient=(0.025,) * 3, shadow_ambient=(0.1,) * 3) -> SceneBlocks:
    groups = dict(spheres_buf=spheres, planes_buf=planes, surfaces_buf=surfaces, boxes_buf=boxes, toruses_buf=toruses, rings_buf=rings,
                  lights_point_buf=lights_point, lights_direct_buf=lights_direct)
    blocks = {k: b"".join(v) for k, v in groups.items()}
    blocks["scene_buf"] = scene_block(width, height, cam_pos, cam_quat, depth)
    defines = (len(spheres), len(planes), len(surfaces), len(boxes), len(toruses), len(rings), len(lights_point), len(lights_direct), depth,
               *ambient, *shadow_ambient)
    return SceneBlocks(defines=defines, blocks=blocks)
