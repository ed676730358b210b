// rtx/GLWrapper.h -- header-only C++ shim with the reference's GLWrapper surface
// (reference: src/GLWrapper.h:17-38) on top of the C ABI in rtx.h.
//
// A main.cpp-style program written against the reference compiles against this header with the
// GLFW loop removed (see examples/demo_main.cpp and INTEGRATION.md): same constructor, same
// init_window / init_shaders / load_cubemap / set_skybox / load_texture / init_buffer /
// update_buffer / draw / enable_SMAA / stop, same print-and-exit error behaviour
// (GLWrapper.cpp:224-227,371-375; utils.h:57-63). What is different, by construction:
//   * the "window" is a device colour target; `window` is a null GLFWwindow* (windowing, input and
//     presentation are outside the replaced path -- SURVEY.md section 8(b),(f));
//   * enable_SMAA(preset) switches on the SMAA post-process (the three passes of GLWrapper.cpp:173-204 as HIP kernels). Its two
//     look-up tables need nothing from the caller: the library computes them (rtx/smaa_tables.h; byte-identical to the reference's
//     AreaTex.h / SearchTex.h arrays). Built inside the reference tree the shim still hands the tree's own arrays over in init_shaders,
//     like SMAA_Builder does; set_SMAA_tables replaces them with any others;
//   * image files: the tracer boundary takes decoded 8-bit texels. load_texture/load_cubemap
//     decode through a pluggable function (set_image_decoder); the built-in decoder reads PNG (png_decode.h), JPEG
//     (jpeg_decode.h; stb_image's arithmetic, so the texels equal the reference's) and binary PPM/PGM (P6/P5) and
//     PAM (P7, RGB_ALPHA). With RTX_WITH_STB_IMAGE defined and stb_image.h on the include path the reference's
//     decoder (stbi_load) is used instead;
//   * extra, non-reference conveniences: load_texture_raw, load_cubemap_raw, read_pixels, save_png (the frame as a file,
//     in place of the window the reference presents it in).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../rtx.h"
#include "scene.h"

#ifdef RTX_WITH_STB_IMAGE
#include <stb_image.h>
#endif
#include "png_decode.h"
#include "jpeg_decode.h"
#include "png_write.h"
#if defined(__has_include)
#if __has_include("AreaTex.h") && __has_include("SearchTex.h")
#include "AreaTex.h"     // the reference's own tables (src/AreaTex.h, src/SearchTex.h), when the shim is built in its tree
#include "SearchTex.h"
#define RTX_SHIM_HAVE_SMAA_TABLES 1
#endif
#endif

#ifndef ASSETS_DIR
#define ASSETS_DIR "."
#endif

// ---- the few GL names scene/main code mentions --------------------------------------------
#ifndef __glad_h_
typedef unsigned int GLuint;
typedef unsigned int GLenum;
#define GL_TEXTURE_2D 0x0DE1
#define GL_REPEAT 0x2901
#define GL_CLAMP_TO_EDGE 0x812F
#define GL_TEXTURE0 0x84C0
#define GL_TEXTURE1 0x84C1
#define GL_TEXTURE2 0x84C2
#define GL_TEXTURE3 0x84C3
#define GL_TEXTURE4 0x84C4
#define GL_TEXTURE5 0x84C5
#define GL_TEXTURE6 0x84C6
#define GL_TEXTURE7 0x84C7
#define GL_TEXTURE_CUBE_MAP 0x8513
#define GL_TRUE 1
#define GL_FALSE 0
#endif
struct GLFWwindow;  // never instantiated

enum SMAA_PRESET { LOW, MEDIUM, HIGH, ULTRA };  // reference: src/SMAA_Builder.h:9-12

namespace rtx_shim {
inline int& active_unit() { static int u = 0; return u; }
inline void die(const char* what)
{
    std::fprintf(stderr, "rtx: %s: %s\n", what, rtx_last_error());
    std::exit(1);
}
inline void check(int status, const char* what) { if (status != RTX_OK) die(what); }

// decoder: path -> malloc'ed interleaved 8-bit texels, row 0 first; returns nullptr on failure
typedef unsigned char* (*image_decoder)(const char* path, int* w, int* h, int* channels);

inline unsigned char* decode_pnm(const char* path, int* w, int* h, int* channels)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) return nullptr;
    char magic[3] = {0, 0, 0};
    unsigned char* out = nullptr;
    int maxv = 0;
    if (std::fscanf(f, "%2s", magic) == 1) {
        if (!std::strcmp(magic, "P6") || !std::strcmp(magic, "P5")) {
            *channels = magic[1] == '6' ? 3 : 1;
            if (std::fscanf(f, "%d %d %d", w, h, &maxv) == 3 && maxv == 255) {
                std::fgetc(f);
                size_t n = static_cast<size_t>(*w) * *h * *channels;
                out = static_cast<unsigned char*>(std::malloc(n));
                if (out && std::fread(out, 1, n, f) != n) { std::free(out); out = nullptr; }
            }
        } else if (!std::strcmp(magic, "P7")) {
            char key[32], val[32];
            *w = *h = *channels = 0;
            while (std::fscanf(f, "%31s", key) == 1 && std::strcmp(key, "ENDHDR")) {
                if (std::fscanf(f, "%31s", val) != 1) break;
                if (!std::strcmp(key, "WIDTH")) *w = std::atoi(val);
                if (!std::strcmp(key, "HEIGHT")) *h = std::atoi(val);
                if (!std::strcmp(key, "DEPTH")) *channels = std::atoi(val);
            }
            std::fgetc(f);
            size_t n = static_cast<size_t>(*w) * *h * *channels;
            if (n) {
                out = static_cast<unsigned char*>(std::malloc(n));
                if (out && std::fread(out, 1, n, f) != n) { std::free(out); out = nullptr; }
            }
        }
    }
    std::fclose(f);
    return out;
}
#ifdef RTX_WITH_STB_IMAGE
inline unsigned char* decode_stb(const char* path, int* w, int* h, int* channels) { return stbi_load(path, w, h, channels, 0); }
#endif
// built-in: PNG (png_decode.h), JPEG (jpeg_decode.h) -- both with stb_image's output convention -- or binary PNM/PAM,
// told apart by the file's first bytes
inline unsigned char* decode_builtin(const char* path, int* w, int* h, int* channels)
{
    unsigned char head[2] = {0, 0};
    FILE* f = std::fopen(path, "rb");
    if (!f) return nullptr;
    const size_t got = std::fread(head, 1, 2, f);
    std::fclose(f);
    if (got == 2 && head[0] == 0x89 && head[1] == 'P') return rtx_png::decode_file(path, w, h, channels);
    if (got == 2 && head[0] == 0xFF && head[1] == 0xD8) return rtx_jpeg::decode_file(path, w, h, channels);
    return decode_pnm(path, w, h, channels);
}
inline image_decoder& decoder()
{
#ifdef RTX_WITH_STB_IMAGE
    static image_decoder d = decode_stb;
#else
    static image_decoder d = decode_builtin;
#endif
    return d;
}
}  // namespace rtx_shim

// main.cpp:178-187 re-binds its textures every frame with raw GL calls; keep those lines compiling.
#ifndef __glad_h_
inline void glActiveTexture(GLenum texture) { rtx_shim::active_unit() = static_cast<int>(texture) - GL_TEXTURE0; }
inline void glBindTexture(GLenum /*target*/, GLuint texture)
{
    rtx_shim::check(rtx_bind_texture(rtx_current(), rtx_shim::active_unit(), texture), "glBindTexture");
}
#endif

class GLWrapper {
public:
    GLWrapper(int width, int height, bool fullScreen)  // GLWrapper.cpp:12-18
        : window(nullptr), ctx(nullptr), width(width), height(height), fullScreen(fullScreen), useCustomResolution(true) {}
    explicit GLWrapper(bool fullScreen)  // GLWrapper.cpp:20-23; "monitor resolution" has no meaning here: 1920x1080
        : window(nullptr), ctx(nullptr), width(1920), height(1080), fullScreen(fullScreen), useCustomResolution(false) {}
    ~GLWrapper() { if (ctx) rtx_destroy(ctx); }
    GLWrapper(const GLWrapper&) = delete;
    GLWrapper& operator=(const GLWrapper&) = delete;

    int getWidth() { return width; }
    int getHeight() { return height; }
    GLuint getProgramId() { return 1; }
    rtx_context* context() { return ctx; }

    bool init_window()  // GLWrapper.cpp:61-133 -> returns false on failure, never exits
    {
        // Which GPUs draw() uses is the environment's choice, so that a main.cpp-style program needs no new code for a multi-GPU node:
        //   RTX_DEVICES = "0,1,2,3" (device ids) or "4" (the first four) -> rtx_create_multi: interleaved row bands, assembled on the first;
        //   RTX_GATHER  = "rccl" (default) | "peer" (hipMemcpyPeerAsync) | "loopback" (rccl incl. the root's own bands: one-GPU diagnostic);   RTX_DEVICE = id of the single device otherwise (default 0).
        std::vector<int> ids;
        if (const char* list = std::getenv("RTX_DEVICES")) {
            const std::string t = list;
            if (t.find(',') == std::string::npos) {
                for (int k = 0; k < std::atoi(t.c_str()); k++) ids.push_back(k);
            } else {
                size_t pos = 0;
                while (pos <= t.size()) {
                    const size_t next = t.find(',', pos);
                    ids.push_back(std::atoi(t.substr(pos, next == std::string::npos ? std::string::npos : next - pos).c_str()));
                    if (next == std::string::npos) break;
                    pos = next + 1;
                }
            }
        }
        int st;
        if (ids.size() > 1) {
            const char* g = std::getenv("RTX_GATHER");
            st = rtx_create_multi(width, height, static_cast<int>(ids.size()), ids.data(), (g && g[0] == 'p') ? RTX_GATHER_PEER_COPY : (g && g[0] == 'l') ? RTX_GATHER_RCCL_LOOPBACK : RTX_GATHER_RCCL, &ctx);
        } else {
            const char* dev = std::getenv("RTX_DEVICE");
            st = rtx_create(width, height, ids.size() == 1 ? ids[0] : (dev ? std::atoi(dev) : 0), &ctx);
        }
        if (st != RTX_OK) {
            std::fprintf(stderr, "rtx_create failed: %s\n", rtx_last_error());
            return false;
        }
        std::printf("rtx %s\n", rtx_version());
        if (SMAA_enabled) rtx_shim::check(rtx_enable_smaa(ctx, static_cast<int>(SMAA_preset)), "enable_SMAA");
        return true;
    }

    void init_shaders(rt_defines& defines)  // GLWrapper.cpp:232-277
    {
        rtx_defines d;
        static_assert(sizeof(rtx_defines) == sizeof(rt_defines), "rt_defines layout");
        std::memcpy(&d, &defines, sizeof d);
        rtx_shim::check(rtx_specialize(ctx, &d), "init_shaders");
#ifdef RTX_SHIM_HAVE_SMAA_TABLES
        if (SMAA_enabled)   // GLWrapper.cpp:251-270: areaTex = load_area_texture(), searchTex = load_search_texture()
            rtx_shim::check(rtx_smaa_set_tables(ctx, areaTexBytes, AREATEX_WIDTH, AREATEX_HEIGHT, searchTexBytes, SEARCHTEX_WIDTH, SEARCHTEX_HEIGHT), "init_shaders (SMAA tables)");
#endif
    }

    void set_skybox(unsigned int textureId)  // GLWrapper.cpp:135-141
    {
        rtx_shim::check(rtx_sampler_unit(ctx, "skybox", 0), "set_skybox");
        rtx_shim::check(rtx_bind_texture(ctx, 0, textureId), "set_skybox");
        rtx_shim::active_unit() = 0;
    }

    void stop() {}                           // GLWrapper.cpp:143-147: nothing to tear down before ~GLWrapper
    void enable_SMAA(SMAA_PRESET preset)     // GLWrapper.cpp:149-153; the reference calls it before init_window (main.cpp:32), here any time works
    {
        SMAA_enabled = true;
        SMAA_preset = preset;
        if (ctx) rtx_shim::check(rtx_enable_smaa(ctx, static_cast<int>(preset)), "enable_SMAA");
    }
    // extra: SMAA_Builder::load_area_texture / load_search_texture with caller-supplied bytes (160 x 560 RG8, 64 x 16 R8)
    void set_SMAA_tables(const unsigned char* area_rg8, const unsigned char* search_r8)
    {
        rtx_shim::check(rtx_smaa_set_tables(ctx, area_rg8, 160, 560, search_r8, 64, 16), "set_SMAA_tables");
    }

    GLFWwindow* window;

    void draw() { rtx_shim::check(rtx_draw(ctx), "draw"); }  // GLWrapper.cpp:155-165

    static void set_image_decoder(rtx_shim::image_decoder d) { rtx_shim::decoder() = d; }

    static GLuint load_cubemap(std::vector<std::string> faces, bool genMipmap = false)  // GLWrapper.cpp:284-317
    {
        unsigned char* data[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        int fw = 0, fh = 0, fc = 0;
        for (unsigned int i = 0; i < faces.size() && i < 6; i++) {
            int w = 0, h = 0, c = 0;
            data[i] = rtx_shim::decoder()(faces[i].c_str(), &w, &h, &c);
            if (!data[i]) { std::printf("Cubemap tex failed to load at path: %s\n", faces[i].c_str()); continue; }
            if (fw == 0) { fw = w; fh = h; fc = c; }
            if (w != fw || h != fh || c != fc || w != h) { std::free(data[i]); data[i] = nullptr; }
        }
        GLuint id = load_cubemap_raw(fw, fc ? fc : 3, data, genMipmap);
        for (auto p : data) std::free(p);
        return id;
    }
    static GLuint load_cubemap_raw(int face_size, int channels, const unsigned char* const faces[6], bool genMipmap = false)
    {
        uint32_t h = 0;
        rtx_shim::check(rtx_cubemap_create(rtx_current(), face_size, channels, faces, genMipmap ? 1 : 0, &h), "load_cubemap");
        return h;
    }

    GLuint load_texture(int texNum, const char* name, const char* uniformName, GLuint wrapMode = GL_REPEAT)  // GLWrapper.cpp:356-363
    {
        const std::string path = ASSETS_DIR "/textures/" + std::string(name);
        int w = 0, h = 0, c = 0;
        unsigned char* data = rtx_shim::decoder()(path.c_str(), &w, &h, &c);
        GLuint id = 0;
        if (data) {
            id = load_texture_raw(texNum, w, h, c, data, uniformName, wrapMode);
            std::free(data);
        } else {
            std::printf("Texture failed to load at path: %s\n", path.c_str());  // GLWrapper.cpp:347-351: message, continue
            rtx_shim::check(rtx_sampler_unit(ctx, uniformName, texNum), "load_texture");
        }
        return id;
    }
    GLuint load_texture_raw(int texNum, int w, int h, int channels, const unsigned char* texels, const char* uniformName,
                            GLuint wrapMode = GL_REPEAT)
    {
        uint32_t id = 0;
        rtx_shim::check(rtx_texture2d_create(ctx, w, h, channels, texels, wrapMode == GL_CLAMP_TO_EDGE ? RTX_WRAP_CLAMP_TO_EDGE : RTX_WRAP_REPEAT, &id),
                        "load_texture");
        rtx_shim::check(rtx_sampler_unit(ctx, uniformName, texNum), "load_texture");
        rtx_shim::check(rtx_bind_texture(ctx, texNum, id), "load_texture");
        return id;
    }

    void init_buffer(GLuint* ubo, const char* name, int bindingPoint, size_t size, void* data) const  // GLWrapper.cpp:365-379
    {
        uint32_t h = 0;
        int st = rtx_block_create(ctx, name, bindingPoint, size, data, &h);
        if (st == RTX_ERR_NAME) { std::fprintf(stderr, "Invalid ubo block name '%s'", name); std::exit(1); }
        rtx_shim::check(st, "init_buffer");
        *ubo = h;
    }
    static void update_buffer(GLuint ubo, size_t size, void* data)  // GLWrapper.cpp:381-386
    {
        rtx_shim::check(rtx_block_update(rtx_current(), ubo, size, data), "update_buffer");
    }

    // glReadPixels stand-in: RTX_RGBA32F -> w*h*4 floats, RTX_RGBA8 -> w*h*4 bytes; row 0 = bottom
    void read_pixels(int format, void* dst, size_t bytes) { rtx_shim::check(rtx_read_pixels(ctx, format, dst, bytes), "read_pixels"); }
    bool save_png(const char* path)   // what the reference's window shows (RGBA8; after SMAA when that is enabled), top row first
    {
        std::vector<unsigned char> rgba(static_cast<size_t>(width) * static_cast<size_t>(height) * 4);
        read_pixels(RTX_SCREEN_RGBA8, rgba.data(), rgba.size());
        return rtx_png::write_file(path, rgba.data(), width, height, 4, /*bottom_up=*/true);
    }

private:
    rtx_context* ctx;
    int width;
    int height;
    bool fullScreen;
    bool useCustomResolution;
    bool SMAA_enabled = false;
    SMAA_PRESET SMAA_preset = ULTRA;
};
