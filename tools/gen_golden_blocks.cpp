// gen_golden_blocks.cpp -- CONTAINER-ONLY fixture generator (needs /root/reference).
//
// Compiles this repo's scene recipes (raytracing_opengl_amd/csrc/host/scene_recipes.h) against
// the REFERENCE's own headers and factory code (src/scene.h, src/Surface.h, src/SceneManager.cpp,
// vendored GLM 0.9.9.7) and dumps the resulting uniform-block bytes as RTXB containers into
// tests/golden/. The fixtures are data (inputs for the tracer), not reference source.
// Build + run: tools/gen_golden_blocks.sh
#include <cfloat>
#include <cstdio>
#include <string>

#include "SceneManager.h"  // reference
#include "Surface.h"       // reference
#include "scene.h"         // reference

#include "scene_blob.h"
#include "scene_recipes.h"

static void dump(const std::string& path, scene_container& sc)
{
    std::vector<unsigned char> blob = scene_blob::serialize(sc);
    FILE* f = fopen(path.c_str(), "wb");
    fwrite(blob.data(), 1, blob.size(), f);
    fclose(f);
    printf("%s: %zu bytes\n", path.c_str(), blob.size());
}

int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : ".";
    {   // default scene, time 0: 1920x1080, depth 4
        scene_container sc = {};
        scene_recipes::anim_slots slot = scene_recipes::build_default(sc, 1920, 1080, 4);
        scene_recipes::animate_default(sc, slot, 0.0f, 0.0f);
        sc.scene.quat_camera_rotation = glm::quat(glm::vec3(glm::radians(-0.0f), glm::radians(0.0f), 0));  // SceneManager.cpp:50, yaw=pitch=0
        dump(dir + "/default_t0_1920x1080_d4.rtxb", sc);
    }
    {   // default scene after one animation step (pins angleAxis, quat *=, vec3 * quat)
        scene_container sc = {};
        scene_recipes::anim_slots slot = scene_recipes::build_default(sc, 640, 480, 1);
        scene_recipes::animate_default(sc, slot, 0.75f, 12.5f);
        sc.scene.quat_camera_rotation = glm::quat(glm::vec3(glm::radians(-10.0f), glm::radians(25.0f), 0));
        dump(dir + "/default_t12.5_640x480_d1.rtxb", sc);
    }
    {
        scene_container sc = {};
        scene_recipes::build_quadric(sc, 3840, 2160, 4);
        sc.scene.quat_camera_rotation = glm::quat(glm::vec3(glm::radians(-0.0f), glm::radians(0.0f), 0));
        dump(dir + "/quadric_3840x2160_d4.rtxb", sc);
    }
    {
        scene_container sc = {};
        scene_recipes::build_torus(sc, 3840, 2160, 6);
        sc.scene.quat_camera_rotation = glm::quat(glm::vec3(glm::radians(-0.0f), glm::radians(0.0f), 0));
        dump(dir + "/torus_3840x2160_d6.rtxb", sc);
    }
    return 0;
}
