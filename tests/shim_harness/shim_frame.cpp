// shim_frame.cpp -- test client of the header-only C++ drop-in surface (include/rtx/GLWrapper.h, SceneManager.h, Surface.h).
//
// A main.cpp-shaped program (reference src/main.cpp:23-195 without GLFW): GLWrapper, enable_SMAA, init_window, scene through
// SceneManager::create_* (scene_recipes.h), init_shaders, load_cubemap / load_texture FROM IMAGE FILES (decoded by the shim's
// own PNG / JPEG readers), SceneManager::init, one update + the per-frame texture binds + draw. It then dumps the frame three
// ways -- RGBA32F raw, RGBA8 raw, save_png (and the SMAA screen when smaa = 1, with tables read from smaa_area.bin / smaa_search.bin) -- for tests/test_gpu_widened.py to compare with the oracle. Run with the current
// directory holding textures/{sky0..5.png,t1.jpg,t2.jpg,t3.jpg,ring.png,box.png} (ASSETS_DIR defaults to ".").
//   shim_frame W H depth time delta yaw pitch smaa(0/1) out_prefix
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "rtx/GLWrapper.h"
#include "rtx/SceneManager.h"
#include "rtx/Surface.h"

#include "../../raytracing_opengl_amd/csrc/host/scene_recipes.h"

static bool dump(const std::string& path, const void* p, size_t n)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(p, 1, n, f) == n;
    std::fclose(f);
    return ok;
}

int main(int argc, char** argv)
{
    if (argc < 10) { std::fprintf(stderr, "usage: shim_frame W H depth time delta yaw pitch smaa out_prefix\n"); return 2; }
    const int W = std::atoi(argv[1]), H = std::atoi(argv[2]), depth = std::atoi(argv[3]);
    const float time = static_cast<float>(std::atof(argv[4])), delta = static_cast<float>(std::atof(argv[5]));
    const float yaw = static_cast<float>(std::atof(argv[6])), pitch = static_cast<float>(std::atof(argv[7]));
    const bool smaa = std::atoi(argv[8]) != 0;
    const std::string out = argv[9];

    GLWrapper glWrapper(W, H, false);
    if (smaa) glWrapper.enable_SMAA(ULTRA);          // main.cpp:32
    if (!glWrapper.init_window()) return 1;
    std::vector<unsigned char> area(160 * 560 * 2), search(64 * 16);
    if (smaa) {   // the tables are not part of this repository: the test supplies them as files
        FILE* fa = std::fopen("smaa_area.bin", "rb");
        FILE* fs = std::fopen("smaa_search.bin", "rb");
        const bool ok_t = fa && fs && std::fread(area.data(), 1, area.size(), fa) == area.size() && std::fread(search.data(), 1, search.size(), fs) == search.size();
        if (fa) std::fclose(fa);
        if (fs) std::fclose(fs);
        if (!ok_t) { std::fprintf(stderr, "smaa_area.bin / smaa_search.bin missing\n"); return 4; }
        glWrapper.set_SMAA_tables(area.data(), search.data());
    }

    scene_container scene = {};
    scene_recipes::anim_slots slots = scene_recipes::build_default(scene, W, H, depth);
    rt_defines defines = scene.get_defines();
    glWrapper.init_shaders(defines);

    std::vector<std::string> faces = {"textures/sky0.png", "textures/sky1.png", "textures/sky2.png",
                                      "textures/sky3.png", "textures/sky4.png", "textures/sky5.png"};
    // main.cpp:137-147 passes the default (false); SHIM_CUBE_MIPS=1: load_cubemap(faces, true), GLWrapper.cpp:307-310
    glWrapper.set_skybox(GLWrapper::load_cubemap(faces, std::getenv("SHIM_CUBE_MIPS") != nullptr));
    auto jupiterTex = glWrapper.load_texture(1, "t1.jpg", "texture_sphere_1");  // main.cpp:149-153
    auto saturnTex = glWrapper.load_texture(2, "t2.jpg", "texture_sphere_2");
    auto marsTex = glWrapper.load_texture(3, "t3.jpg", "texture_sphere_3");
    auto ringTex = glWrapper.load_texture(4, "ring.png", "texture_ring");
    auto boxTex = glWrapper.load_texture(5, "box.png", "texture_box");

    SceneManager scene_manager(W, H, &scene, &glWrapper);
    scene_manager.init();
    scene_manager.set_view(yaw, pitch);

    scene_recipes::animate_default(scene, slots, delta, time);   // update_scene(), main.cpp:197-246
    scene_manager.update(delta);
    glActiveTexture(GL_TEXTURE1); glBindTexture(GL_TEXTURE_2D, jupiterTex);   // main.cpp:178-187
    glActiveTexture(GL_TEXTURE2); glBindTexture(GL_TEXTURE_2D, saturnTex);
    glActiveTexture(GL_TEXTURE3); glBindTexture(GL_TEXTURE_2D, marsTex);
    glActiveTexture(GL_TEXTURE4); glBindTexture(GL_TEXTURE_2D, ringTex);
    glActiveTexture(GL_TEXTURE5); glBindTexture(GL_TEXTURE_2D, boxTex);
    glWrapper.draw();

    const size_t px = static_cast<size_t>(W) * H;
    std::vector<float> f32(px * 4);
    std::vector<unsigned char> u8(px * 4);
    glWrapper.read_pixels(RTX_RGBA32F, f32.data(), f32.size() * sizeof(float));
    glWrapper.read_pixels(RTX_RGBA8, u8.data(), u8.size());
    bool ok = dump(out + ".f32", f32.data(), f32.size() * sizeof(float)) && dump(out + ".u8", u8.data(), u8.size());
    if (smaa) {
        std::vector<unsigned char> screen(px * 4);
        glWrapper.read_pixels(RTX_SCREEN_RGBA8, screen.data(), screen.size());
        ok = dump(out + ".screen", screen.data(), screen.size()) && ok;
    }
    ok = glWrapper.save_png((out + ".png").c_str()) && ok;
    glWrapper.stop();
    return ok ? 0 : 3;
}
