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

struct rt_defines {  // reference: src/scene.h:7-20 -- the tracer's specialisation key
    int sphere_size;
    int plane_size;
    int surface_size;
    int box_size;
    int torus_size;
    int ring_size;
    int light_point_size;
    int light_direct_size;
    int iterations;
    glm::vec3 ambient_color;
    glm::vec3 shadow_ambient;
};

typedef struct {  // 64 B, reference: src/scene.h:22-35
    glm::vec3 color; float __p1;
    glm::vec3 absorb;
    float diffuse;
    float reflect;
    float refract;
    int specular;
    float kd;
    float ks;
    float __padding[3];
} rt_material;

typedef struct {  // 112 B, reference: src/scene.h:37-44
    rt_material material;
    glm::vec4 obj;  // centre xyz + radius
    glm::quat quat_rotation = glm::quat(1, 0, 0, 0);
    int textureNum;
    bool hollow;
    unsigned char __p0[3] = {0, 0, 0};
    float __padding[2];
} rt_sphere;

typedef struct {  // 96 B, reference: src/scene.h:46-50
    rt_material material;
    glm::vec3 pos; float __p1;
    glm::vec3 normal; float __p2;
} rt_plane;

typedef struct {  // 112 B, reference: src/scene.h:52-58
    rt_material mat;
    glm::quat quat_rotation = glm::quat(1, 0, 0, 0);
    glm::vec3 pos; float __p1;
    glm::vec3 form;  // half extents
    int textureNum;
} rt_box;

typedef struct {  // 112 B, reference: src/scene.h:60-65
    rt_material mat;
    glm::quat quat_rotation = glm::quat(1, 0, 0, 0);
    glm::vec3 pos; float __p1;
    glm::vec2 form;  // x = major radius, y = tube radius
    float __p2[2];
} rt_torus;

typedef struct {  // 112 B, reference: src/scene.h:67-73
    rt_material mat;
    glm::quat quat_rotation = glm::quat(1, 0, 0, 0);
    glm::vec3 pos; int textureNum;
    float r1, r2;  // SQUARED inner / outer radius
    float __p2[2];
} rt_ring;

typedef struct {  // 160 B, reference: src/scene.h:75-95
    rt_material mat;
    glm::quat quat_rotation = glm::quat(1, 0, 0, 0);
    float xMin = -FLT_MAX;  // clip box, WORLD space
    float yMin = -FLT_MAX;
    float zMin = -FLT_MAX;
    float __p0;
    float xMax = FLT_MAX;
    float yMax = FLT_MAX;
    float zMax = FLT_MAX;
    float __p1;
    glm::vec3 pos;
    float a;  // x^2
    float b;  // y^2
    float c;  // z^2
    float d;  // z
    float e;  // y
    float f;  // const
    float __padding[3];
} rt_surface;

typedef enum { sphere, light } primitiveType;

struct rt_light_direct {  // 32 B, reference: src/scene.h:99-104
    glm::vec3 direction; float __p1;
    glm::vec3 color;
    float intensity;
};

struct rt_light_point {  // 48 B, reference: src/scene.h:106-114
    glm::vec4 pos;  // xyz + radius of the visible light sphere
    glm::vec3 color;
    float intensity;
    float linear_k;
    float quadratic_k;
    float __padding[2];
};

typedef struct {  // 64 B, reference: src/scene.h:116-126
    glm::quat quat_camera_rotation;
    glm::vec3 camera_pos; float __p1;
    glm::vec3 bg_color;
    int canvas_width;
    int canvas_height;
    int reflect_depth;
    float __padding[2];
} rt_scene;

struct scene_container {  // reference: src/scene.h:128-154
    rt_scene scene;
    glm::vec3 ambient_color;
    glm::vec3 shadow_ambient;
    std::vector<rt_sphere> spheres;
    std::vector<rt_plane> planes;
    std::vector<rt_surface> surfaces;
    std::vector<rt_box> boxes;
    std::vector<rt_torus> toruses;
    std::vector<rt_ring> rings;
    std::vector<rt_light_point> lights_point;
    std::vector<rt_light_direct> lights_direct;

    rt_defines get_defines()
    {
        rt_defines d;
        d.sphere_size = static_cast<int>(spheres.size());
        d.plane_size = static_cast<int>(planes.size());
        d.surface_size = static_cast<int>(surfaces.size());
        d.box_size = static_cast<int>(boxes.size());
        d.torus_size = static_cast<int>(toruses.size());
        d.ring_size = static_cast<int>(rings.size());
        d.light_point_size = static_cast<int>(lights_point.size());
        d.light_direct_size = static_cast<int>(lights_direct.size());
        d.iterations = scene.reflect_depth;
        d.ambient_color = ambient_color;
        d.shadow_ambient = shadow_ambient;
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
