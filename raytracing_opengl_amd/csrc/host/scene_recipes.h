// scene_recipes.h -- the three bench/parity scenes, written ONLY against the reference's public
// scene-description API (scene.h PODs, SceneManager::create_*, SurfaceFactory, glm::vec/quat).
//
// The same file is compiled twice:
//   * by the product's host library against include/rtx/*.h (this repo's headers), and
//   * by tools/gen_golden_blocks.sh against the reference's own src/*.h + vendored GLM, which is
//     how tests/golden/*.bin are produced. Byte-equality of the two is the drop-in proof for the
//     scene-description half of the boundary (tests/test_scene_blocks.py).
//
// Scenes
//   default : the scene every BASELINE config names -- reference src/main.cpp:43-132, frozen by
//             animate_default(time, delta) = reference update_scene (main.cpp:197-246)
//   quadric : 96 quadrics, 6 README types x 16   (seed 3)
//   torus   : 64 tori on an 8x8 grid              (seed 4)
// Include after scene.h / Surface.h / SceneManager.h of whichever header set is in use.
#pragma once

#include <cmath>

namespace scene_recipes {

struct anim_slots {  // indices of the animated primitives (main.cpp:12-19)
    int jupiter, saturn, saturn_ring, mars, crate, torus;
};

inline glm::quat saturn_tilt() { return glm::quat(glm::vec3(glm::radians(15.f), 0, 0)); }  // main.cpp:21

inline void common_camera_and_lights(scene_container& sc, int canvas_w, int canvas_h, int depth)
{
    sc.scene = SceneManager::create_scene(canvas_w, canvas_h);
    sc.scene.camera_pos = {0, 0, -5};
    sc.scene.reflect_depth = depth;
    sc.shadow_ambient = glm::vec3{0.1, 0.1, 0.1};
    sc.ambient_color = glm::vec3{0.025, 0.025, 0.025};
    sc.lights_point.push_back(SceneManager::create_light_point({3, 5, 0, 0.1}, {1, 1, 1}, 25.5));
    sc.lights_direct.push_back(SceneManager::create_light_direct({3, -1, 1}, {1, 1, 1}, 1.5));
}

// ---- default scene ------------------------------------------------------------------------
inline anim_slots build_default(scene_container& sc, int canvas_w, int canvas_h, int depth)
{
    anim_slots slot = {-1, -1, -1, -1, -1, -1};
    common_camera_and_lights(sc, canvas_w, canvas_h, depth);

    // three unit spheres: blue mirror-ish, red hollow, glass (refract 1.125, absorbing)
    sc.spheres.push_back(SceneManager::create_sphere({2, 0, 6}, 1, SceneManager::create_material({0, 0, 1}, 50, 0.35)));
    sc.spheres.push_back(SceneManager::create_sphere({-1, 0, 6}, 1, SceneManager::create_material({1, 0, 0}, 100, 0.1), true));
    sc.spheres.push_back(
        SceneManager::create_sphere({0.5, 2, 6}, 1, SceneManager::create_material({1, 1, 1}, 200, 0.1, 1.125, {1, 0, 2}, 1), true));

    // planets: textured, matte black material (colour comes from the texture)
    const rt_material planet_mat = SceneManager::create_material({}, 0, 0.0f);
    const int saturn_radius = 4150;
    struct { float radius; int tex; int* slot; } planets[] = {
        {5000, 1, &slot.jupiter}, {static_cast<float>(saturn_radius), 2, &slot.saturn}, {500, 3, &slot.mars}};
    for (auto& p : planets) {
        rt_sphere s = SceneManager::create_sphere({}, p.radius, planet_mat);
        s.textureNum = p.tex;
        if (p.tex == 2) s.quat_rotation = saturn_tilt();
        sc.spheres.push_back(s);
        *p.slot = static_cast<int>(sc.spheres.size()) - 1;
    }
    {
        rt_ring ring = SceneManager::create_ring({}, saturn_radius * 1.1166, saturn_radius * 2.35, SceneManager::create_material({}, 0, 0));
        ring.textureNum = 4;
        ring.quat_rotation = glm::angleAxis(glm::radians(90.f), glm::vec3(1, 0, 0)) * saturn_tilt();
        sc.rings.push_back(ring);
        slot.saturn_ring = static_cast<int>(sc.rings.size()) - 1;
    }

    // floor slab and textured crate
    sc.boxes.push_back(SceneManager::create_box({0, -1.2, 6}, {10, 0.2, 5}, SceneManager::create_material({1, 0.6, 0}, 100, 0.05)));
    rt_box crate = SceneManager::create_box({8, 1, 6}, {1, 1, 1}, SceneManager::create_material({0.8, 0.7, 0}, 50, 0.0));
    crate.textureNum = 5;
    sc.boxes.push_back(crate);
    slot.crate = static_cast<int>(sc.boxes.size()) - 1;

    rt_torus torus = SceneManager::create_torus({-9, 0.5, 6}, {1.0, 0.5}, SceneManager::create_material({0.5, 0.4, 1}, 200, 0.2));
    torus.quat_rotation = glm::quat(glm::vec3(glm::radians(45.f), 0, 0));
    sc.toruses.push_back(torus);
    slot.torus = static_cast<int>(sc.toruses.size()) - 1;

    rt_surface cone = SurfaceFactory::GetEllipticCone(1 / 3.0f, 1 / 3.0f, 1,
                                                      SceneManager::create_material({234 / 255.0f, 17 / 255.0f, 82 / 255.0f}, 200, 0.2));
    cone.pos = {-5, 4, 6};
    cone.quat_rotation = glm::quat(glm::vec3(glm::radians(90.f), 0, 0));
    cone.yMin = -1;
    cone.yMax = 4;
    sc.surfaces.push_back(cone);

    rt_surface cylinder = SurfaceFactory::GetEllipticCylinder(
        1 / 2.0f, 1 / 2.0f, SceneManager::create_material({200 / 255.0f, 255 / 255.0f, 0 / 255.0f}, 200, 0.2));
    cylinder.pos = {5, 0, 6};
    cylinder.quat_rotation = glm::quat(glm::vec3(glm::radians(90.f), 0, 0));
    cylinder.yMin = -1;
    cylinder.yMax = 1;
    sc.surfaces.push_back(cylinder);
    return slot;
}

// main.cpp calls unqualified cos()/sin() on float arguments with only <cmath> in scope, which
// (libstdc++) resolves to the C double functions; the product is then formed in double and
// narrowed on assignment. Spelled out here so the choice does not depend on include order.
inline double orbit_cos(float a) { return ::cos(static_cast<double>(a)); }
inline double orbit_sin(float a) { return ::sin(static_cast<double>(a)); }

// reference update_scene (main.cpp:197-246): orbits + spins as a pure function of (time, delta)
inline void animate_default(scene_container& sc, const anim_slots& slot, float deltaTime, float time)
{
    if (slot.jupiter != -1) {
        rt_sphere* j = &sc.spheres[slot.jupiter];
        const float jupiterSpeed = 0.02;
        j->obj.x = orbit_cos(time * jupiterSpeed) * 20000;
        j->obj.z = orbit_sin(time * jupiterSpeed) * 20000;
        j->quat_rotation *= glm::angleAxis(deltaTime / 15, glm::vec3(0, 1, 0));
    }
    if (slot.saturn != -1 && slot.saturn_ring != -1) {
        rt_sphere* s = &sc.spheres[slot.saturn];
        rt_ring* ring = &sc.rings[slot.saturn_ring];
        const float speed = 0.0082;
        const float dist = 35000;
        const float offset = 1;
        s->obj.x = orbit_cos(time * speed + offset) * dist;
        s->obj.z = orbit_sin(time * speed + offset) * dist;
        glm::vec3 axis = glm::vec3(0, 1, 0) * saturn_tilt();
        s->quat_rotation *= glm::angleAxis(deltaTime / 10, axis);
        ring->pos.x = orbit_cos(time * speed + offset) * dist;
        ring->pos.z = orbit_sin(time * speed + offset) * dist;
    }
    if (slot.mars != -1) {
        rt_sphere* m = &sc.spheres[slot.mars];
        const float marsSpeed = 0.05;
        m->obj.x = orbit_cos(time * marsSpeed + 0.5f) * 10000;
        m->obj.z = orbit_sin(time * marsSpeed + 0.5f) * 10000;
        m->obj.y = -orbit_cos(time * marsSpeed) * 3000;
        m->quat_rotation *= glm::angleAxis(deltaTime / 5, glm::vec3(0, 1, 0));
    }
    if (slot.crate != -1) {
        sc.boxes[slot.crate].quat_rotation *= glm::angleAxis(deltaTime, glm::vec3(0.5774, 0.5774, 0.5774));
    }
    if (slot.torus != -1) {
        sc.toruses[slot.torus].quat_rotation *= glm::angleAxis(deltaTime, glm::vec3(0, 1, 0));
    }
}

// ---- synthetic stress scenes ----------------------------------------
struct lcg32 {  // x = x*1664525 + 1013904223 (mod 2^32); u = (x >> 8) / 2^24
    unsigned int x;
    float next()
    {
        x = x * 1664525u + 1013904223u;
        return static_cast<float>(x >> 8) / 16777216.0f;
    }
};

inline void add_floor_plane(scene_container& sc)
{
    sc.planes.push_back(SceneManager::create_plane({0, 1, 0}, {0, -12, 0}, SceneManager::create_material({0.6, 0.6, 0.6}, 50, 0.1)));
}

inline void build_quadric(scene_container& sc, int canvas_w, int canvas_h, int depth)
{
    common_camera_and_lights(sc, canvas_w, canvas_h, depth);
    add_floor_plane(sc);
    lcg32 rng = {3u};
    const float two_pi = 6.28318530717958647692f;
    for (int row = 0; row < 6; row++) {
        for (int col = 0; col < 16; col++) {
            const float zj = rng.next();
            const float ex = rng.next() * two_pi, ey = rng.next() * two_pi, ez = rng.next() * two_pi;
            const float cr = 0.2f + 0.8f * rng.next(), cg = 0.2f + 0.8f * rng.next(), cb = 0.2f + 0.8f * rng.next();
            const rt_material m = SceneManager::create_material({cr, cg, cb}, 100, (col % 2 == 0) ? 0.2f : 0.0f);
            rt_surface s;
            switch (row) {
                case 0: s = SurfaceFactory::GetEllipsoid(0.8f, 0.8f, 1.1f, m); break;
                case 1: s = SurfaceFactory::GetEllipticCone(0.5f, 0.5f, 1.0f, m); break;
                case 2: s = SurfaceFactory::GetEllipticCylinder(0.5f, 0.5f, m); break;
                case 3: s = SurfaceFactory::GetEllipticParaboloid(0.8f, 0.8f, m); break;
                case 4: s = SurfaceFactory::GetHyperbolicParaboloid(0.8f, 0.8f, m); break;
                default: s = SurfaceFactory::GetEllipticHyperboloidOneSheet(0.5f, 0.5f, 1.0f, m); break;
            }
            s.pos = {-19.5f + 2.6f * col, -9.0f + 3.6f * row, 20.0f + 2.0f * zj};
            s.quat_rotation = glm::quat(glm::vec3(ex, ey, ez));
            s.xMin = s.pos.x - 1.2f; s.xMax = s.pos.x + 1.2f;
            s.yMin = s.pos.y - 1.2f; s.yMax = s.pos.y + 1.2f;
            s.zMin = s.pos.z - 1.2f; s.zMax = s.pos.z + 1.2f;
            sc.surfaces.push_back(s);
        }
    }
}

inline void build_torus(scene_container& sc, int canvas_w, int canvas_h, int depth)
{
    common_camera_and_lights(sc, canvas_w, canvas_h, depth);
    add_floor_plane(sc);
    lcg32 rng = {4u};
    const float two_pi = 6.28318530717958647692f;
    for (int row = 0; row < 8; row++) {
        for (int col = 0; col < 8; col++) {
            const float zj = rng.next();
            const float ex = rng.next() * two_pi, ey = rng.next() * two_pi, ez = rng.next() * two_pi;
            const float cr = 0.2f + 0.8f * rng.next(), cg = 0.2f + 0.8f * rng.next(), cb = 0.2f + 0.8f * rng.next();
            const rt_material m = SceneManager::create_material({cr, cg, cb}, 200, ((row + col) % 2 == 0) ? 0.2f : 0.0f);
            rt_torus t = SceneManager::create_torus({-14.7f + 4.2f * col, -8.4f + 2.4f * row, 16.0f + 2.0f * zj}, {0.9f, 0.3f}, m);
            t.quat_rotation = glm::quat(glm::vec3(ex, ey, ez));
            sc.toruses.push_back(t);
        }
    }
}

}  // namespace scene_recipes
