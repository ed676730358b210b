// rtx/scene.h -- scene-description PODs, source-compatible with the reference's src/scene.h.
//
// Every struct below is the host mirror of one std140 record the tracer consumes
// (reference: src/scene.h:22-126 <-> assets/shaders/rt.frag:24-113). Field names, order and
// byte offsets are part of the drop-in boundary: SceneManager hands `std::vector<T>::data()`
// straight to GLWrapper::init_buffer / update_buffer (reference: src/SceneManager.cpp:238-276),
// so the bytes ARE the wire format. Offsets are pinned by the static_asserts at the end of this
// file and by the golden block dumps in tests/golden/.
//
// Differences from the reference header (none visible to scene code):
//   * <cfloat> is included here (the reference forgets it and relies on include order);
//   * implicit padding is spelled out so that `T x = {};` leaves deterministic bytes.
#pragma once

#include <cfloat>
#include <cstddef>
#include <vector>

#include "glm_compat.h"

// One line per 16-byte std140 slot: the layout reads off the page.

struct rt_defines {  // reference: src/scene.h:7-20 -- the tracer's specialisation key (nine ints, two colours = 60 B)
    int sphere_size, plane_size, surface_size, box_size, torus_size, ring_size, light_point_size, light_direct_size;
    int iterations;
    glm::vec3 ambient_color, shadow_ambient;
};

struct rt_material {  // 64 B, reference: src/scene.h:22-35
    glm::vec3 color;  float __p1;
    glm::vec3 absorb; float diffuse;
    float reflect, refract; int specular; float kd;
    float ks, __padding[3];
};

struct rt_sphere {  // 112 B, reference: src/scene.h:37-44
    rt_material material;
    glm::vec4 obj;                                      // centre xyz + radius
    glm::quat quat_rotation = glm::quat(1, 0, 0, 0);    // rotates the texture lookup only
    int textureNum; bool hollow; unsigned char __p0[3] = {0, 0, 0}; float __padding[2];
};

struct rt_plane {  // 96 B, reference: src/scene.h:46-50
    rt_material material;
    glm::vec3 pos;    float __p1;
    glm::vec3 normal; float __p2;
};

struct rt_box {  // 112 B, reference: src/scene.h:52-58
    rt_material mat;
    glm::quat quat_rotation = glm::quat(1, 0, 0, 0);
    glm::vec3 pos;  float __p1;
    glm::vec3 form; int textureNum;                     // half extents
};

struct rt_torus {  // 112 B, reference: src/scene.h:60-65
    rt_material mat;
    glm::quat quat_rotation = glm::quat(1, 0, 0, 0);
    glm::vec3 pos;  float __p1;
    glm::vec2 form; float __p2[2];                      // x = major radius, y = tube radius
};

struct rt_ring {  // 112 B, reference: src/scene.h:67-73
    rt_material mat;
    glm::quat quat_rotation = glm::quat(1, 0, 0, 0);
    glm::vec3 pos; int textureNum;
    float r1, r2, __p2[2];                              // r1, r2: SQUARED inner / outer radius
};

struct rt_surface {  // 160 B, reference: src/scene.h:75-95
    rt_material mat;
    glm::quat quat_rotation = glm::quat(1, 0, 0, 0);
    float xMin = -FLT_MAX, yMin = -FLT_MAX, zMin = -FLT_MAX, __p0;   // clip box, WORLD space
    float xMax = FLT_MAX, yMax = FLT_MAX, zMax = FLT_MAX, __p1;
    glm::vec3 pos; float a;                             // a x^2 + b y^2 + c z^2 + d z + e y + f = 0 in the local frame
    float b, c, d, e;
    float f, __padding[3];
};

enum primitiveType { sphere, light };

struct rt_light_direct {  // 32 B, reference: src/scene.h:99-104
    glm::vec3 direction; float __p1;
    glm::vec3 color;     float intensity;
};

struct rt_light_point {  // 48 B, reference: src/scene.h:106-114
    glm::vec4 pos;                                      // xyz + radius of the visible light sphere
    glm::vec3 color; float intensity;
    float linear_k, quadratic_k, __padding[2];
};

struct rt_scene {  // 64 B, reference: src/scene.h:116-126
    glm::quat quat_camera_rotation;
    glm::vec3 camera_pos; float __p1;
    glm::vec3 bg_color;   int canvas_width;
    int canvas_height, reflect_depth; float __padding[2];
};

struct scene_container {  // reference: src/scene.h:128-154
    rt_scene scene;
    glm::vec3 ambient_color, shadow_ambient;
    std::vector<rt_sphere> spheres;
    std::vector<rt_plane> planes;
    std::vector<rt_surface> surfaces;
    std::vector<rt_box> boxes;
    std::vector<rt_torus> toruses;
    std::vector<rt_ring> rings;
    std::vector<rt_light_point> lights_point;
    std::vector<rt_light_direct> lights_direct;

    rt_defines get_defines()   // array sizes + bounce depth + the two colours: what the tracer is specialised on
    {
        auto n = [](size_t k) { return static_cast<int>(k); };
        rt_defines d = {n(spheres.size()), n(planes.size()), n(surfaces.size()), n(boxes.size()), n(toruses.size()), n(rings.size()),
                        n(lights_point.size()), n(lights_direct.size()), scene.reflect_depth, ambient_color, shadow_ambient};
        return d;
    }
};

// ---- std140 layout pins -------------------------------------------
#define RTX_PIN(T, field, off) static_assert(offsetof(T, field) == (off), #T "." #field " offset")
static_assert(sizeof(rt_material) == 64, "rt_material");
RTX_PIN(rt_material, absorb, 16); RTX_PIN(rt_material, diffuse, 28); RTX_PIN(rt_material, reflect, 32);
RTX_PIN(rt_material, refract, 36); RTX_PIN(rt_material, specular, 40); RTX_PIN(rt_material, kd, 44);
RTX_PIN(rt_material, ks, 48);
static_assert(sizeof(rt_sphere) == 112, "rt_sphere");
RTX_PIN(rt_sphere, obj, 64); RTX_PIN(rt_sphere, quat_rotation, 80); RTX_PIN(rt_sphere, textureNum, 96);
RTX_PIN(rt_sphere, hollow, 100);
static_assert(sizeof(rt_plane) == 96, "rt_plane");
RTX_PIN(rt_plane, pos, 64); RTX_PIN(rt_plane, normal, 80);
static_assert(sizeof(rt_box) == 112, "rt_box");
RTX_PIN(rt_box, quat_rotation, 64); RTX_PIN(rt_box, pos, 80); RTX_PIN(rt_box, form, 96); RTX_PIN(rt_box, textureNum, 108);
static_assert(sizeof(rt_torus) == 112, "rt_torus");
RTX_PIN(rt_torus, quat_rotation, 64); RTX_PIN(rt_torus, pos, 80); RTX_PIN(rt_torus, form, 96);
static_assert(sizeof(rt_ring) == 112, "rt_ring");
RTX_PIN(rt_ring, quat_rotation, 64); RTX_PIN(rt_ring, pos, 80); RTX_PIN(rt_ring, textureNum, 92);
RTX_PIN(rt_ring, r1, 96); RTX_PIN(rt_ring, r2, 100);
static_assert(sizeof(rt_surface) == 160, "rt_surface");
RTX_PIN(rt_surface, quat_rotation, 64); RTX_PIN(rt_surface, xMin, 80); RTX_PIN(rt_surface, xMax, 96);
RTX_PIN(rt_surface, pos, 112); RTX_PIN(rt_surface, a, 124); RTX_PIN(rt_surface, f, 144);
static_assert(sizeof(rt_light_direct) == 32, "rt_light_direct");
RTX_PIN(rt_light_direct, color, 16); RTX_PIN(rt_light_direct, intensity, 28);
static_assert(sizeof(rt_light_point) == 48, "rt_light_point");
RTX_PIN(rt_light_point, color, 16); RTX_PIN(rt_light_point, intensity, 28); RTX_PIN(rt_light_point, linear_k, 32);
RTX_PIN(rt_light_point, quadratic_k, 36);
static_assert(sizeof(rt_scene) == 64, "rt_scene");
RTX_PIN(rt_scene, camera_pos, 16); RTX_PIN(rt_scene, bg_color, 32); RTX_PIN(rt_scene, canvas_width, 44);
RTX_PIN(rt_scene, canvas_height, 48); RTX_PIN(rt_scene, reflect_depth, 52);
#undef RTX_PIN
