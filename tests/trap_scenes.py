"""Scenes that exercise the reference traps no BASELINE config covers:
total internal reflection, refractive boxes (mirror term + i--), ray origins inside boxes (T21),
exact-zero direction components (box NaN path T5), light-sphere hits, planes (one-sided, no shadows),
hollow spheres, the degenerate quadric branch (T4), multiple textured rings in a shadow ray (T10)."""
import math

from scene_util import (box, light_direct, light_point, make_scene, material, plane, quat_euler, ring, sphere, surface, torus)


def glass_and_tir(w, h, depth=6):
    """Glass slab + glass box seen at grazing angles: rays inside the box meet faces beyond the critical
    angle (TIR break), refraction index < 1 from outside (TIR on entry), absorb, mirror terms."""
    glass = material((1, 1, 1), 200, 0.08, 1.5, (0.6, 0.1, 0.3), 1.0)
    thin = material((0.9, 1, 0.9), 100, 0.15, 0.7, (0.1, 0.1, 0.1), 1.0)   # refract < 1: FresnelReflectAmount n1 > n2 outside
    return make_scene(
        w, h, depth,
        spheres=[sphere((-2.2, 0.3, 5.0), 0.9, material((0.2, 0.4, 1.0), 80, 0.4)), sphere((2.4, 0.2, 7.5), 1.0, thin, hollow=True)],
        planes=[plane((0, 1, 0), (0, -1.5, 0), material((0.7, 0.7, 0.6), 30, 0.2))],
        boxes=[box((0.2, -0.2, 4.5), (1.2, 1.0, 0.8), glass, quat=quat_euler(0.2, 0.6, 0.1)),
               box((3.0, -0.9, 4.0), (0.6, 0.6, 0.6), material((1, 0.5, 0.1), 60, 0.0))],
        lights_point=[light_point((2.0, 4.0, 1.0), 0.3)], lights_direct=[light_direct((1, -2, 1))],
        cam_pos=(0.0, 0.6, -1.5))


def camera_inside_box_and_axis_rays(w, h, depth=3):
    """Camera INSIDE a big box (negative entry distance accepted, T21) looking exactly along +z with an
    odd-sized canvas offset so that some rays have exact-zero direction components (box NaN path, T5)."""
    room = material((0.6, 0.7, 0.9), 20, 0.1)
    return make_scene(
        w, h, depth,
        spheres=[sphere((0.0, 0.0, 4.0), 1.0, material((1, 0.2, 0.2), 100, 0.3)), sphere((0.0, 3.0, 2.0), 0.11, material((1, 1, 0), 10, 0))],
        boxes=[box((0, 0, 0), (6, 4, 8), room), box((-2.0, -1.0, 3.0), (0.7, 0.7, 0.7), material((0.2, 0.9, 0.3), 50, 0.2), quat=quat_euler(0, math.pi / 2, 0))],
        toruses=[torus((2.2, -0.5, 3.5), 0.8, 0.3, material((0.9, 0.6, 0.1), 120, 0.25), quat=quat_euler(1.1, 0.3, 0))],
        lights_point=[light_point((0.0, 3.0, 2.0), 0.4)],   # big light sphere: closest-hit light hits (main + side rays)
        cam_pos=(0.0, 0.0, -2.0))


def quadric_degenerate_and_rings(w, h, depth=4):
    """Unrotated cone and cylinder whose asymptotic / axis directions coincide with camera rays (T4 branch),
    hyperbolic paraboloid with an unbounded clip box (no cull possible), and TWO textured rings crossed by
    the same shadow rays (alpha accumulation order, T10) plus an untextured ring."""
    q90 = quat_euler(math.pi / 2, 0, 0)
    return make_scene(
        w, h, depth,
        spheres=[sphere((0.0, -0.3, 6.0), 0.7, material((0.9, 0.9, 0.9), 150, 0.5))],
        planes=[plane((0, 1, 0), (0, -2.0, 0), material((0.5, 0.55, 0.6), 40, 0.15))],
        surfaces=[surface((-2.5, 1.0, 7.0), material((0.9, 0.1, 0.4), 200, 0.2), a=1, b=1, c=-1, vmin=(-9, -1.5, 3), vmax=(9, 3.5, 11)),  # cone along z: rays on its surface directions
                  surface((0.5, 0.2, 7.0), material((0.8, 1, 0), 200, 0.2), a=0.02, b=0.02, f=-1, vmin=(-9, -1.5, 3), vmax=(9, 6.5, 14)),   # wide cylinder along the view axis: |p2| < 1e-6 for the central rays
                  surface((0.0, 2.5, 9.0), material((0.1, 0.7, 0.9), 90, 0.1), a=0.5, b=-0.5, d=-1, quat=q90)],                            # saddle, unbounded clip
        rings=[ring((0.0, 1.2, 5.0), 0.6, 1.6, material((0, 0, 0), 0, 0), quat=quat_euler(0.3, 0.2, 0), texture=4),
               ring((0.3, 2.0, 5.2), 0.4, 1.9, material((0, 0, 0), 0, 0), quat=quat_euler(0.25, -0.2, 0), texture=4),
               ring((-1.5, 0.5, 4.5), 0.3, 0.8, material((0.9, 0.8, 0.2), 60, 0.0), quat=quat_euler(0.9, 0.0, 0.4))],
        lights_point=[light_point((0.5, 5.0, 4.5), 0.1)], lights_direct=[light_direct((0.2, -1, 0.3))],
        cam_pos=(0.0, 0.5, -2.0))


def planes_lights_and_glass_spheres(w, h, depth=5):
    """Well-conditioned refraction (two glass spheres in a row: i-- without the box pathology, T2; absorbDistance carried
    from the first sphere into the second, T12; TIR inside), planes seen from the front and from behind (one-sided and not
    shadowing, T1), a hollow sphere that shadows like a solid one (T8), a big point-light sphere in view and in mirror
    rays (closest-hit only, never an occluder; getReflectedColor's light branch, T3), an opaque ring casting a shadow."""
    glass = material((1, 1, 1), 120, 0.1, 1.33, (0.35, 0.05, 0.2), 1.0)
    dense = material((1, 1, 1), 60, 0.05, 1.7, (0.05, 0.3, 0.3), 1.0)
    return make_scene(
        w, h, depth,
        spheres=[sphere((-0.9, 0.2, 4.0), 0.8, glass), sphere((-0.7, 0.3, 6.5), 1.0, dense),
                 sphere((1.8, 0.0, 5.0), 0.9, material((0.9, 0.3, 0.2), 40, 0.0), hollow=True),
                 sphere((3.2, -0.6, 3.6), 0.5, material((0.2, 0.8, 0.3), 90, 0.6))],
        planes=[plane((0, 1, 0), (0, -1.2, 0), material((0.7, 0.7, 0.75), 25, 0.25)),      # floor, seen from above
                plane((0, 1, 0), (0, 3.0, 0), material((1.0, 0.1, 0.1), 10, 0.0)),         # "ceiling" facing up: seen from below -> invisible
                plane((0.3, 0.2, -1), (0, 0, 12.0), material((0.3, 0.4, 0.8), 15, 0.1))],  # back wall, un-normalised normal
        rings=[ring((2.6, 1.6, 4.6), 0.3, 0.9, material((0.9, 0.8, 0.1), 30, 0.0), quat=quat_euler(1.2, 0.3, 0.0))],
        lights_point=[light_point((0.6, 2.2, 3.2), 0.35)], lights_direct=[light_direct((-0.4, -1, 0.5))],
        cam_pos=(0.2, 0.4, -2.5))


ALL = {"glass_tir": glass_and_tir, "inside_box": camera_inside_box_and_axis_rays, "degenerate_rings": quadric_degenerate_and_rings,
       "planes_glass": planes_lights_and_glass_spheres}
