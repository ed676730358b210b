// demo_main.cpp -- a main.cpp-style program on the drop-in C++ surface (include/rtx/*.h).
//
// Structure follows the reference's src/main.cpp: construct GLWrapper, init_window, build the
// scene with SceneManager::create_* / SurfaceFactory, init_shaders(defines), skybox + textures,
// SceneManager::init(), then the frame loop (update_scene -> scene_manager.update -> bind
// textures -> draw). What differs from the reference program is only what lies outside the
// replaced path: no GLFW window/vsync/input (a fixed number of frames driven by a synthetic
// clock), and the last frame is written as a PNG (demo_frame.png). Textures: procedural by default, so that the demo needs no files;
// built with -DDEMO_ASSET_FILES -DASSETS_DIR=\"<reference checkout>/assets\" it loads the reference's own JPEG / PNG
// files through load_cubemap / load_texture exactly as main.cpp:137-153 does (decoded by include/rtx/jpeg_decode.h and
// png_decode.h to the same texels stb_image gives the reference).
//
// Build (see examples/Makefile):  g++ -std=c++17 -Iinclude examples/demo_main.cpp -L... -lrtx_hip
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rtx/GLWrapper.h"
#include "rtx/SceneManager.h"
#include "rtx/Surface.h"

#include "../raytracing_opengl_amd/csrc/host/scene_recipes.h"

static int wind_width = 1280;   // reference main.cpp:7-8
static int wind_height = 720;

// tiny procedural textures so the demo needs no asset files
[[maybe_unused]] static std::vector<unsigned char> checker(int w, int h, int c, int cell, unsigned char lo, unsigned char hi)
{
    std::vector<unsigned char> t(static_cast<size_t>(w) * h * c);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int k = 0; k < c; k++)
                t[(static_cast<size_t>(y) * w + x) * c + k] = (k == 3) ? (((x / cell) % 3) ? 255 : 90) : ((((x / cell) + (y / cell)) & 1) ? hi : lo) / (1 + k % 3);
    return t;
}

int main(int argc, char** argv)
{
    const int frames = argc > 1 ? std::atoi(argv[1]) : 60;
    GLWrapper glWrapper(wind_width, wind_height, false);
    glWrapper.enable_SMAA(ULTRA);  // main.cpp:32, unconditionally like the reference: the library computes the two look-up tables itself
                                   // (include/rtx/smaa_tables.h, byte-identical to the reference's AreaTex.h / SearchTex.h arrays)
    if (!glWrapper.init_window()) return 1;
    wind_width = glWrapper.getWidth();
    wind_height = glWrapper.getHeight();
    if (wind_width % 2 == 1) wind_width++;   // main.cpp:39-41
    if (wind_height % 2 == 1) wind_height++;

    scene_container scene = {};
    scene_recipes::anim_slots slots = scene_recipes::build_default(scene, wind_width, wind_height, 5);

    rt_defines defines = scene.get_defines();
    glWrapper.init_shaders(defines);

#ifdef DEMO_ASSET_FILES
    std::vector<std::string> faces = {   // main.cpp:137-145
        ASSETS_DIR "/textures/sb_nebula/GalaxyTex_PositiveX.jpg", ASSETS_DIR "/textures/sb_nebula/GalaxyTex_NegativeX.jpg",
        ASSETS_DIR "/textures/sb_nebula/GalaxyTex_PositiveY.jpg", ASSETS_DIR "/textures/sb_nebula/GalaxyTex_NegativeY.jpg",
        ASSETS_DIR "/textures/sb_nebula/GalaxyTex_PositiveZ.jpg", ASSETS_DIR "/textures/sb_nebula/GalaxyTex_NegativeZ.jpg"};
    glWrapper.set_skybox(GLWrapper::load_cubemap(faces, false));
    auto jupiterTex = glWrapper.load_texture(1, "8k_jupiter.jpg", "texture_sphere_1");
    auto saturnTex = glWrapper.load_texture(2, "8k_saturn.jpg", "texture_sphere_2");
    auto marsTex = glWrapper.load_texture(3, "2k_mars.jpg", "texture_sphere_3");
    auto ringTex = glWrapper.load_texture(4, "8k_saturn_ring_alpha.png", "texture_ring");
    auto boxTex = glWrapper.load_texture(5, "container.png", "texture_box");
#else
    std::vector<std::vector<unsigned char>> faces;
    const unsigned char* face_ptr[6];
    for (int f = 0; f < 6; f++) { faces.push_back(checker(256, 256, 3, 32, 5, 40 + 20 * f)); face_ptr[f] = faces.back().data(); }
    glWrapper.set_skybox(GLWrapper::load_cubemap_raw(256, 3, face_ptr, false));

    auto tj = checker(1024, 512, 3, 32, 120, 220), tsat = checker(1024, 512, 3, 16, 150, 230), tm = checker(512, 256, 3, 16, 90, 200);
    auto tr = checker(2048, 125, 4, 8, 140, 210), tb = checker(128, 128, 4, 16, 100, 180);
    auto jupiterTex = glWrapper.load_texture_raw(1, 1024, 512, 3, tj.data(), "texture_sphere_1");
    auto saturnTex = glWrapper.load_texture_raw(2, 1024, 512, 3, tsat.data(), "texture_sphere_2");
    auto marsTex = glWrapper.load_texture_raw(3, 512, 256, 3, tm.data(), "texture_sphere_3");
    auto ringTex = glWrapper.load_texture_raw(4, 2048, 125, 4, tr.data(), "texture_ring");
    auto boxTex = glWrapper.load_texture_raw(5, 128, 128, 4, tb.data(), "texture_box");
#endif

    SceneManager scene_manager(wind_width, wind_height, &scene, &glWrapper);
    scene_manager.init();

    float currentTime = 0.0f;
    const float deltaTime = 1.0f / 60.0f;  // synthetic clock instead of glfwGetTime()
    const auto t0 = std::chrono::steady_clock::now();
    for (int frame = 0; frame < frames; frame++) {
        currentTime += deltaTime;
        scene_recipes::animate_default(scene, slots, deltaTime, currentTime);  // = update_scene(), main.cpp:197-246
        scene_manager.update(deltaTime);
        glActiveTexture(GL_TEXTURE1); glBindTexture(GL_TEXTURE_2D, jupiterTex);
        glActiveTexture(GL_TEXTURE2); glBindTexture(GL_TEXTURE_2D, saturnTex);
        glActiveTexture(GL_TEXTURE3); glBindTexture(GL_TEXTURE_2D, marsTex);
        glActiveTexture(GL_TEXTURE4); glBindTexture(GL_TEXTURE_2D, ringTex);
        glActiveTexture(GL_TEXTURE5); glBindTexture(GL_TEXTURE_2D, boxTex);
        glWrapper.draw();
    }
    std::vector<unsigned char> rgba(static_cast<size_t>(glWrapper.getWidth()) * glWrapper.getHeight() * 4);
    glWrapper.read_pixels(RTX_SCREEN_RGBA8, rgba.data(), rgba.size());  // synchronises; what the reference's window shows: the frame after SMAA
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("FPS: %.1f (%d frames, %dx%d, depth 5)\n", frames / secs, frames, glWrapper.getWidth(), glWrapper.getHeight());

    if (!rtx_png::write_file("demo_frame.png", rgba.data(), glWrapper.getWidth(), glWrapper.getHeight(), 4, /*bottom_up=*/true))
        std::fprintf(stderr, "could not write demo_frame.png\n");
    glWrapper.stop();
    return 0;
}
