// scene_blob.h -- flat "RTXB" container for one scene_container: what SceneManager would push
// through GLWrapper::init_buffer, in binding-point order (SceneManager.cpp:244-255).
//
//   u32 magic 'RTXB' | u32 sizes[9] | rt_defines (60 B) | block bytes, binding 0..8, unpadded
//
// Used by the host library (rtxh_scene_build), by tools/gen_golden_blocks.cpp (compiled against
// the REFERENCE headers) and parsed by raytracing_opengl_amd/scenes.py.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace scene_blob {

static const uint32_t kMagic = 0x42585452u;  // "RTXB"
static const char* const kBlockNames[9] = {"scene_buf", "spheres_buf", "planes_buf", "surfaces_buf", "boxes_buf",
                                           "toruses_buf", "rings_buf", "lights_point_buf", "lights_direct_buf"};

template <class T>
inline void append(std::vector<unsigned char>& out, const T* p, size_t n)
{
    const unsigned char* b = reinterpret_cast<const unsigned char*>(p);
    out.insert(out.end(), b, b + n * sizeof(T));
}

inline std::vector<unsigned char> serialize(scene_container& sc)
{
    std::vector<unsigned char> out;
    const uint32_t sizes[9] = {
        static_cast<uint32_t>(sizeof(rt_scene)),
        static_cast<uint32_t>(sc.spheres.size() * sizeof(rt_sphere)),
        static_cast<uint32_t>(sc.planes.size() * sizeof(rt_plane)),
        static_cast<uint32_t>(sc.surfaces.size() * sizeof(rt_surface)),
        static_cast<uint32_t>(sc.boxes.size() * sizeof(rt_box)),
        static_cast<uint32_t>(sc.toruses.size() * sizeof(rt_torus)),
        static_cast<uint32_t>(sc.rings.size() * sizeof(rt_ring)),
        static_cast<uint32_t>(sc.lights_point.size() * sizeof(rt_light_point)),
        static_cast<uint32_t>(sc.lights_direct.size() * sizeof(rt_light_direct))};
    append(out, &kMagic, 1);
    append(out, sizes, 9);
    rt_defines d = sc.get_defines();
    append(out, &d, 1);
    append(out, &sc.scene, 1);
    append(out, sc.spheres.data(), sc.spheres.size());
    append(out, sc.planes.data(), sc.planes.size());
    append(out, sc.surfaces.data(), sc.surfaces.size());
    append(out, sc.boxes.data(), sc.boxes.size());
    append(out, sc.toruses.data(), sc.toruses.size());
    append(out, sc.rings.data(), sc.rings.size());
    append(out, sc.lights_point.data(), sc.lights_point.size());
    append(out, sc.lights_direct.data(), sc.lights_direct.size());
    return out;
}

}  // namespace scene_blob
