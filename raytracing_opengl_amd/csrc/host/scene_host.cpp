// scene_host.cpp -- host-side scene construction exported with a C ABI (librtx_host.so).
//
// Builds the bench/parity scenes with this repo's scene-description headers (include/rtx/*.h:
// the SceneManager / SurfaceFactory / scene.h surface of the reference) and returns them as an
// RTXB container (scene_blob.h). No device code here: the library is used by the Python host
// layer, the tests and bench.py to obtain the exact uniform-block bytes a main.cpp-style program
// would upload.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rtx/SceneManager.h"
#include "rtx/Surface.h"
#include "rtx/scene.h"

#include "scene_blob.h"
#include "scene_recipes.h"

extern "C" {

// kind: "default" | "quadric" | "torus". time/delta drive the default scene's animation
// (reference main.cpp:197-246); yaw/pitch (degrees) and cam_pos (NULL = recipe default) set the
// camera exactly like SceneManager::update_scene (SceneManager.cpp:43-50).
// Returns the number of bytes the container needs; copies min(needed, cap) bytes into out.
__attribute__((visibility("default"))) size_t rtxh_scene_build(const char* kind, int canvas_w, int canvas_h, int depth, float time,
                                                               float delta, float yaw_deg, float pitch_deg, const float* cam_pos,
                                                               void* out, size_t cap)
{
    scene_container sc = {};
    const std::string k = kind ? kind : "";
    if (k == "default") {
        scene_recipes::anim_slots slot = scene_recipes::build_default(sc, canvas_w, canvas_h, depth);
        scene_recipes::animate_default(sc, slot, delta, time);
    } else if (k == "quadric") {
        scene_recipes::build_quadric(sc, canvas_w, canvas_h, depth);
    } else if (k == "torus") {
        scene_recipes::build_torus(sc, canvas_w, canvas_h, depth);
    } else {
        return 0;
    }
    if (cam_pos) sc.scene.camera_pos = glm::vec3(cam_pos[0], cam_pos[1], cam_pos[2]);
    sc.scene.quat_camera_rotation = glm::quat(glm::vec3(glm::radians(-pitch_deg), glm::radians(yaw_deg), 0));
    std::vector<unsigned char> blob = scene_blob::serialize(sc);
    if (out && cap) std::memcpy(out, blob.data(), blob.size() < cap ? blob.size() : cap);
    return blob.size();
}

// The shim's built-in image decoder (PNG / JPEG / PNM), for tests and Python callers: returns the byte count w*h*channels
// (0 on failure) and copies min(needed, cap) bytes of interleaved 8-bit texels, row 0 = top row of the file.
__attribute__((visibility("default"))) size_t rtxh_decode_image(const char* path, int* w, int* h, int* channels, void* out, size_t cap)
{
    unsigned char* px = rtx_shim::decode_builtin(path, w, h, channels);
    if (!px) return 0;
    const size_t need = static_cast<size_t>(*w) * static_cast<size_t>(*h) * static_cast<size_t>(*channels);
    if (out && cap) std::memcpy(out, px, need < cap ? need : cap);
    std::free(px);
    return need;
}

// include/rtx/png_write.h for tests: 1 on success
__attribute__((visibility("default"))) int rtxh_write_png(const char* path, const unsigned char* pixels, int w, int h, int channels, int bottom_up)
{
    return rtx_png::write_file(path, pixels, w, h, channels, bottom_up != 0) ? 1 : 0;
}

__attribute__((visibility("default"))) const char* rtxh_block_name(int binding)
{
    return (binding >= 0 && binding < 9) ? scene_blob::kBlockNames[binding] : nullptr;
}

}  // extern "C"
