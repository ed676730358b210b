/* glref.c -- TEST INFRASTRUCTURE, build container only.
 *
 * Runs the REFERENCE's own fragment shader (assets/shaders/rt.frag, read from /root/reference at run time,
 * never copied into this repository) on Mesa's llvmpipe software rasteriser, head-less: the DRI software
 * driver (swrast_dri.so) is loaded directly through its loader interface (GL/internal/dri_interface.h), so no
 * X server or EGL is needed. The frames it produces pin oracle/rt_oracle.c against the reference itself
 * (tools/gen_reference_frames.py -> tests/golden/ref_frame_*.npz).
 *
 * What this file does is what the reference's GLWrapper does around the shader (GLWrapper.cpp:61-133 window and
 * quad, :232-277 "{NAME}" templating of rt.frag, :284-363 textures, :365-386 uniform blocks, :155-165 draw), with
 * an RGBA32F colour attachment instead of the window so that the un-quantised FragColor can be read back.
 * Nothing here is part of the product: only tools/ and tests/ load the library, and only in this container.
 */
#define GL_GLEXT_PROTOTYPES 0
#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/internal/dri_interface.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char g_err[4096];
const char* glref_error(void) { return g_err; }
#define FAIL(...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return -1; } while (0)

/* ---- swrast loader callbacks: the window-system drawable is never looked at (we render to an FBO) ---- */
static void get_drawable_info(__DRIdrawable* d, int* x, int* y, int* w, int* h, void* p) { (void)d; (void)p; *x = *y = 0; *w = *h = 16; }
static void put_image(__DRIdrawable* d, int op, int x, int y, int w, int h, char* data, void* p) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void get_image(__DRIdrawable* d, int x, int y, int w, int h, char* data, void* p) { (void)d; (void)x; (void)y; (void)p; memset(data, 0, (size_t)w * h * 4); }
static void put_image2(__DRIdrawable* d, int op, int x, int y, int w, int h, int stride, char* data, void* p) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }
static void get_image2(__DRIdrawable* d, int x, int y, int w, int h, int stride, char* data, void* p) { (void)d; (void)x; (void)y; (void)w; (void)p; memset(data, 0, (size_t)stride * h); }
static const __DRIswrastLoaderExtension g_loader = {
    .base = {__DRI_SWRAST_LOADER, 3},
    .getDrawableInfo = get_drawable_info, .putImage = put_image, .getImage = get_image, .putImage2 = put_image2, .getImage2 = get_image2,
};
static const __DRIextension* g_loader_exts[] = {&g_loader.base, NULL};

static const __DRIcoreExtension* g_core;
static const __DRIswrastExtension* g_swrast;
static __DRIscreen* g_screen;
static __DRIcontext* g_ctx;
static __DRIdrawable* g_draw;
static void* (*g_getproc)(const char*);

#define GLFN(ret, name, ...) static ret (*p_##name)(__VA_ARGS__)
GLFN(const GLubyte*, glGetString, GLenum);
GLFN(GLenum, glGetError, void);
GLFN(GLuint, glCreateShader, GLenum);
GLFN(void, glShaderSource, GLuint, GLsizei, const GLchar* const*, const GLint*);
GLFN(void, glCompileShader, GLuint);
GLFN(void, glGetShaderiv, GLuint, GLenum, GLint*);
GLFN(void, glGetShaderInfoLog, GLuint, GLsizei, GLsizei*, GLchar*);
GLFN(GLuint, glCreateProgram, void);
GLFN(void, glAttachShader, GLuint, GLuint);
GLFN(void, glLinkProgram, GLuint);
GLFN(void, glGetProgramiv, GLuint, GLenum, GLint*);
GLFN(void, glGetProgramInfoLog, GLuint, GLsizei, GLsizei*, GLchar*);
GLFN(void, glUseProgram, GLuint);
GLFN(void, glDeleteProgram, GLuint);
GLFN(void, glDeleteShader, GLuint);
GLFN(GLuint, glGetUniformBlockIndex, GLuint, const GLchar*);
GLFN(void, glUniformBlockBinding, GLuint, GLuint, GLuint);
GLFN(GLint, glGetUniformLocation, GLuint, const GLchar*);
GLFN(void, glUniform1i, GLint, GLint);
GLFN(void, glGenBuffers, GLsizei, GLuint*);
GLFN(void, glDeleteBuffers, GLsizei, const GLuint*);
GLFN(void, glBindBuffer, GLenum, GLuint);
GLFN(void, glBufferData, GLenum, GLsizeiptr, const void*, GLenum);
GLFN(void, glBindBufferBase, GLenum, GLuint, GLuint);
GLFN(void, glGenVertexArrays, GLsizei, GLuint*);
GLFN(void, glBindVertexArray, GLuint);
GLFN(void, glEnableVertexAttribArray, GLuint);
GLFN(void, glVertexAttribPointer, GLuint, GLint, GLenum, GLboolean, GLsizei, const void*);
GLFN(void, glGenTextures, GLsizei, GLuint*);
GLFN(void, glDeleteTextures, GLsizei, const GLuint*);
GLFN(void, glBindTexture, GLenum, GLuint);
GLFN(void, glActiveTexture, GLenum);
GLFN(void, glTexImage2D, GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void*);
GLFN(void, glTexParameteri, GLenum, GLenum, GLint);
GLFN(void, glGenerateMipmap, GLenum);
GLFN(void, glPixelStorei, GLenum, GLint);
GLFN(void, glGenFramebuffers, GLsizei, GLuint*);
GLFN(void, glDeleteFramebuffers, GLsizei, const GLuint*);
GLFN(void, glBindFramebuffer, GLenum, GLuint);
GLFN(void, glFramebufferTexture2D, GLenum, GLenum, GLenum, GLuint, GLint);
GLFN(GLenum, glCheckFramebufferStatus, GLenum);
GLFN(void, glViewport, GLint, GLint, GLsizei, GLsizei);
GLFN(void, glClearColor, GLfloat, GLfloat, GLfloat, GLfloat);
GLFN(void, glClear, GLbitfield);
GLFN(void, glDrawArrays, GLenum, GLint, GLsizei);
GLFN(void, glReadPixels, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, void*);
GLFN(void, glFinish, void);
GLFN(void, glDisable, GLenum);
GLFN(void, glGetTexImage, GLenum, GLint, GLenum, GLenum, void*);

static int load_gl(void)
{
#define L(name) do { *(void**)(&p_##name) = g_getproc(#name); if (!p_##name) FAIL("GL entry point %s missing", #name); } while (0)
    L(glGetString); L(glGetError); L(glCreateShader); L(glShaderSource); L(glCompileShader); L(glGetShaderiv); L(glGetShaderInfoLog);
    L(glCreateProgram); L(glAttachShader); L(glLinkProgram); L(glGetProgramiv); L(glGetProgramInfoLog); L(glUseProgram); L(glDeleteProgram);
    L(glDeleteShader); L(glGetUniformBlockIndex); L(glUniformBlockBinding); L(glGetUniformLocation); L(glUniform1i); L(glGenBuffers);
    L(glDeleteBuffers); L(glBindBuffer); L(glBufferData); L(glBindBufferBase); L(glGenVertexArrays); L(glBindVertexArray);
    L(glEnableVertexAttribArray); L(glVertexAttribPointer); L(glGenTextures); L(glDeleteTextures); L(glBindTexture); L(glActiveTexture);
    L(glTexImage2D); L(glTexParameteri); L(glGenerateMipmap); L(glPixelStorei); L(glGenFramebuffers); L(glDeleteFramebuffers);
    L(glBindFramebuffer); L(glFramebufferTexture2D); L(glCheckFramebufferStatus); L(glViewport); L(glClearColor); L(glClear);
    L(glDrawArrays); L(glReadPixels); L(glFinish); L(glDisable); L(glGetTexImage);
#undef L
    return 0;
}

/* Create the head-less GL 3.3 core context on llvmpipe. Returns 0, or -1 (glref_error()). */
int glref_init(const char* dri_dir)
{
    if (g_ctx) return 0;
    char path[1024];
    snprintf(path, sizeof path, "%s/swrast_dri.so", dri_dir && *dri_dir ? dri_dir : "/usr/lib/x86_64-linux-gnu/dri");
    void* api = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!api) FAIL("dlopen libglapi.so.0: %s", dlerror());
    void* drv = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!drv) FAIL("dlopen %s: %s", path, dlerror());
    *(void**)(&g_getproc) = dlsym(api, "_glapi_get_proc_address");
    if (!g_getproc) FAIL("_glapi_get_proc_address not exported by libglapi");
    const __DRIextension** (*get_exts)(void) = NULL;
    *(void**)(&get_exts) = dlsym(drv, "__driDriverGetExtensions_swrast");
    if (!get_exts) FAIL("__driDriverGetExtensions_swrast not found in %s", path);
    const __DRIextension** exts = get_exts();
    for (int i = 0; exts[i]; i++) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) g_core = (const __DRIcoreExtension*)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) g_swrast = (const __DRIswrastExtension*)exts[i];
    }
    if (!g_core || !g_swrast) FAIL("driver lacks DRI_Core / DRI_SWRast");
    const __DRIconfig** configs = NULL;
    if (g_swrast->base.version >= 4) g_screen = g_swrast->createNewScreen2(0, g_loader_exts, exts, &configs, NULL);
    else g_screen = g_swrast->createNewScreen(0, g_loader_exts, &configs, NULL);
    if (!g_screen || !configs || !configs[0]) FAIL("createNewScreen failed");
    const __DRIconfig* cfg = configs[0];
    for (int i = 0; configs[i]; i++) {  /* first RGBA8888 config */
        unsigned r = 0, a = 0, db = 0;
        g_core->getConfigAttrib(configs[i], __DRI_ATTRIB_RED_SIZE, &r);
        g_core->getConfigAttrib(configs[i], __DRI_ATTRIB_ALPHA_SIZE, &a);
        g_core->getConfigAttrib(configs[i], __DRI_ATTRIB_DOUBLE_BUFFER, &db);
        if (r == 8 && a == 8 && !db) { cfg = configs[i]; break; }
    }
    unsigned err = 0;
    const uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 3, __DRI_CTX_ATTRIB_MINOR_VERSION, 3};
    g_ctx = g_swrast->createContextAttribs(g_screen, __DRI_API_OPENGL_CORE, cfg, NULL, 2, attribs, &err, NULL);
    if (!g_ctx) FAIL("createContextAttribs(GL 3.3 core) failed, error %u", err);
    g_draw = g_swrast->createNewDrawable(g_screen, cfg, NULL);
    if (!g_draw) FAIL("createNewDrawable failed");
    if (!g_core->bindContext(g_ctx, g_draw, g_draw)) FAIL("bindContext failed");
    if (load_gl()) return -1;
    return 0;
}

const char* glref_renderer(void) { return g_ctx ? (const char*)p_glGetString(GL_RENDERER) : ""; }
const char* glref_version(void) { return g_ctx ? (const char*)p_glGetString(GL_VERSION) : ""; }

/* ------------------------------------------------------------------------------------------------------------
 * one frame of the reference's program
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct glref_tex2d {
    const char* uniform_name;  /* texture_sphere_1 ... texture_box */
    int unit;                  /* GL_TEXTURE0 + unit, also the sampler's value (GLWrapper.cpp:356-361) */
    int width, height, channels;
    int clamp_to_edge;         /* 0: GL_REPEAT */
    const unsigned char* texels;
    /* diagnostic only: n_levels > 0 uploads these RGBA8 levels (level 0 first, sizes halved with floor) instead of
     * texels + glGenerateMipmap, to separate "which mip texels" from "which level and weights" when comparing samplers */
    int n_levels;
    const unsigned char* const* levels;
} glref_tex2d;
typedef struct glref_cube {
    int size, channels, gen_mipmap;
    const unsigned char* faces[6];  /* +X -X +Y -Y +Z -Z; NULL = face left undefined like a failed load */
} glref_cube;

static GLuint compile(GLenum kind, const char* src, char* log, size_t log_len)
{
    GLuint s = p_glCreateShader(kind);
    p_glShaderSource(s, 1, &src, NULL);
    p_glCompileShader(s);
    GLint ok = 0;
    p_glGetShaderiv(s, GL_COMPILE_STATUS, &ok);
    if (!ok) { p_glGetShaderInfoLog(s, (GLsizei)log_len, NULL, log); return 0; }
    return s;
}

/* vert_src / frag_src: the reference's shaders, the fragment shader already templated by the caller.
 * blocks: n_blocks uniform blocks, block k bound to binding point k (SceneManager.cpp:244-256).
 * out: w*h RGBA float, row 0 = bottom row (glReadPixels order). inactive_mask: bit k set if block k is not an
 * active block of the linked program (the reference would exit there, GLWrapper.cpp:371-375). */
int glref_render(const char* vert_src, const char* frag_src, int w, int h, int n_blocks, const char* const* names,
                 const void* const* data, const size_t* sizes, int n_tex, const glref_tex2d* tex, const glref_cube* cube,
                 float* out, unsigned* inactive_mask)
{
    if (!g_ctx) FAIL("glref_init first");
    char log[3000] = "";
    GLuint vs = compile(GL_VERTEX_SHADER, vert_src, log, sizeof log);
    if (!vs) FAIL("vertex shader: %s", log);
    GLuint fs = compile(GL_FRAGMENT_SHADER, frag_src, log, sizeof log);
    if (!fs) FAIL("fragment shader: %s", log);
    GLuint prog = p_glCreateProgram();
    p_glAttachShader(prog, vs);
    p_glAttachShader(prog, fs);
    p_glLinkProgram(prog);
    GLint ok = 0;
    p_glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) { p_glGetProgramInfoLog(prog, sizeof log, NULL, log); FAIL("link: %s", log); }
    p_glUseProgram(prog);

    GLuint ubo[16] = {0};
    if (n_blocks > 16) FAIL("too many blocks");
    if (inactive_mask) *inactive_mask = 0;
    p_glGenBuffers(n_blocks, ubo);
    for (int k = 0; k < n_blocks; k++) {
        const GLuint idx = p_glGetUniformBlockIndex(prog, names[k]);
        if (idx == GL_INVALID_INDEX) { if (inactive_mask) *inactive_mask |= 1u << k; continue; }
        /* a block declared with one dummy element (count 0) still needs backing store */
        const size_t bytes = sizes[k] ? sizes[k] : 256;
        void* zero = sizes[k] ? NULL : calloc(1, bytes);
        p_glBindBuffer(GL_UNIFORM_BUFFER, ubo[k]);
        p_glBufferData(GL_UNIFORM_BUFFER, (GLsizeiptr)bytes, sizes[k] ? data[k] : zero, GL_DYNAMIC_DRAW);
        free(zero);
        p_glUniformBlockBinding(prog, idx, (GLuint)k);
        p_glBindBufferBase(GL_UNIFORM_BUFFER, (GLuint)k, ubo[k]);
    }
    p_glBindBuffer(GL_UNIFORM_BUFFER, 0);

    GLuint texid[16] = {0}, cubeid = 0;
    if (n_tex > 16) FAIL("too many textures");
    if (cube) {  /* GLWrapper.cpp:284-317 + set_skybox :135-141 (unit 0, sampler "skybox") */
        p_glGenTextures(1, &cubeid);
        p_glBindTexture(GL_TEXTURE_CUBE_MAP, cubeid);
        const GLenum fmt = cube->channels == 4 ? GL_RGBA : (cube->channels == 1 ? GL_RED : GL_RGB);
        for (int f = 0; f < 6; f++)
            if (cube->faces[f]) p_glTexImage2D(GL_TEXTURE_CUBE_MAP_POSITIVE_X + f, 0, GL_RGB, cube->size, cube->size, 0, fmt, GL_UNSIGNED_BYTE, cube->faces[f]);
        if (cube->gen_mipmap) p_glGenerateMipmap(GL_TEXTURE_CUBE_MAP);
        p_glTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_MIN_FILTER, cube->gen_mipmap ? GL_LINEAR_MIPMAP_LINEAR : GL_LINEAR);
        p_glTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
        p_glTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
        p_glTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
        p_glTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_WRAP_R, GL_CLAMP_TO_EDGE);
        p_glUniform1i(p_glGetUniformLocation(prog, "skybox"), 0);
        p_glActiveTexture(GL_TEXTURE0);
        p_glBindTexture(GL_TEXTURE_CUBE_MAP, cubeid);
    }
    p_glGenTextures(n_tex, texid);
    for (int k = 0; k < n_tex; k++) {  /* GLWrapper.cpp:319-363 */
        const GLenum fmt = tex[k].channels == 1 ? GL_RED : (tex[k].channels == 3 ? GL_RGB : GL_RGBA);
        p_glActiveTexture(GL_TEXTURE0 + tex[k].unit);
        p_glBindTexture(GL_TEXTURE_2D, texid[k]);
        if (tex[k].n_levels > 0) {
            int lw = tex[k].width, lh = tex[k].height;
            p_glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
            for (int L = 0; L < tex[k].n_levels; L++) {
                p_glTexImage2D(GL_TEXTURE_2D, L, GL_RGBA8, lw, lh, 0, GL_RGBA, GL_UNSIGNED_BYTE, tex[k].levels[L]);
                lw = lw > 1 ? lw / 2 : 1; lh = lh > 1 ? lh / 2 : 1;
            }
            p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAX_LEVEL, tex[k].n_levels - 1);
            p_glPixelStorei(GL_UNPACK_ALIGNMENT, 4);
        } else {
            p_glTexImage2D(GL_TEXTURE_2D, 0, (GLint)fmt, tex[k].width, tex[k].height, 0, fmt, GL_UNSIGNED_BYTE, tex[k].texels);
            p_glGenerateMipmap(GL_TEXTURE_2D);
        }
        const GLint wrap = tex[k].clamp_to_edge ? GL_CLAMP_TO_EDGE : GL_REPEAT;
        p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, wrap);
        p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, wrap);
        p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR_MIPMAP_LINEAR);
        p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
        p_glUniform1i(p_glGetUniformLocation(prog, tex[k].uniform_name), tex[k].unit);
    }

    /* colour target: RGBA32F so that FragColor comes back un-quantised */
    GLuint fbo = 0, color = 0, vao = 0, vbo = 0;
    p_glActiveTexture(GL_TEXTURE0 + 15);
    p_glGenTextures(1, &color);
    p_glBindTexture(GL_TEXTURE_2D, color);
    p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, w, h, 0, GL_RGBA, GL_FLOAT, NULL);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    p_glGenFramebuffers(1, &fbo);
    p_glBindFramebuffer(GL_FRAMEBUFFER, fbo);
    p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, color, 0);
    if (p_glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) FAIL("framebuffer incomplete");
    p_glBindTexture(GL_TEXTURE_2D, 0);

    /* the full-screen quad: two triangles over [-1,1]^2 with texture coordinates */
    static const float quad[] = {-1, -1, 0, 0, 1, -1, 1, 0, 1, 1, 1, 1, -1, -1, 0, 0, 1, 1, 1, 1, -1, 1, 0, 1};
    p_glGenVertexArrays(1, &vao);
    p_glGenBuffers(1, &vbo);
    p_glBindVertexArray(vao);
    p_glBindBuffer(GL_ARRAY_BUFFER, vbo);
    p_glBufferData(GL_ARRAY_BUFFER, sizeof quad, quad, GL_STATIC_DRAW);
    p_glEnableVertexAttribArray(0);
    p_glVertexAttribPointer(0, 2, GL_FLOAT, GL_FALSE, 4 * sizeof(float), (void*)0);
    p_glEnableVertexAttribArray(1);
    p_glVertexAttribPointer(1, 2, GL_FLOAT, GL_FALSE, 4 * sizeof(float), (void*)(2 * sizeof(float)));

    p_glViewport(0, 0, w, h);
    p_glDisable(GL_DEPTH_TEST);
    p_glDisable(GL_BLEND);
    p_glClearColor(0, 0, 0, 0);
    p_glClear(GL_COLOR_BUFFER_BIT);
    p_glDrawArrays(GL_TRIANGLES, 0, 6);
    p_glFinish();
    p_glPixelStorei(GL_PACK_ALIGNMENT, 1);
    p_glReadPixels(0, 0, w, h, GL_RGBA, GL_FLOAT, out);
    const GLenum e = p_glGetError();

    p_glBindFramebuffer(GL_FRAMEBUFFER, 0);
    p_glBindVertexArray(0);
    p_glDeleteFramebuffers(1, &fbo);
    p_glDeleteTextures(1, &color);
    p_glDeleteTextures(n_tex, texid);
    if (cubeid) p_glDeleteTextures(1, &cubeid);
    p_glDeleteBuffers(n_blocks, ubo);
    p_glDeleteBuffers(1, &vbo);
    p_glUseProgram(0);
    p_glDeleteProgram(prog);
    p_glDeleteShader(vs);
    p_glDeleteShader(fs);
    if (e != GL_NO_ERROR) FAIL("GL error 0x%x", e);
    return 0;
}

/* The mip level `level` that THIS GL implementation's glGenerateMipmap builds for a texture uploaded like the reference uploads it
 * (GLWrapper.cpp:331-337), as RGBA8 into out. Returns width << 16 | height of the level, 0 if it does not exist, -1 on error.
 * (glGenerateMipmap's filter is the implementation's choice; the fixtures store llvmpipe's levels so that the plain reference frames can
 * be compared with the same mip texels on both sides.) */
int glref_generated_mip(int width, int height, int channels, const unsigned char* texels, int level, unsigned char* out)
{
    if (!g_ctx) FAIL("glref_init first");
    int lw = width, lh = height;
    for (int L = 0; L < level; L++) {
        if (lw == 1 && lh == 1) return 0;
        lw = lw > 1 ? lw / 2 : 1; lh = lh > 1 ? lh / 2 : 1;
    }
    GLuint id = 0;
    const GLenum fmt = channels == 1 ? GL_RED : (channels == 3 ? GL_RGB : GL_RGBA);
    p_glActiveTexture(GL_TEXTURE0 + 14);
    p_glGenTextures(1, &id);
    p_glBindTexture(GL_TEXTURE_2D, id);
    p_glTexImage2D(GL_TEXTURE_2D, 0, (GLint)fmt, width, height, 0, fmt, GL_UNSIGNED_BYTE, texels);
    p_glGenerateMipmap(GL_TEXTURE_2D);
    p_glPixelStorei(GL_PACK_ALIGNMENT, 1);
    p_glGetTexImage(GL_TEXTURE_2D, level, GL_RGBA, GL_UNSIGNED_BYTE, out);
    const GLenum e = p_glGetError();
    p_glBindTexture(GL_TEXTURE_2D, 0);
    p_glDeleteTextures(1, &id);
    if (e != GL_NO_ERROR) FAIL("GL error 0x%x", e);
    return (lw << 16) | lh;
}

/* ------------------------------------------------------------------------------------------------------------
 * one generic full-screen pass: the reference's post-process (SMAA) programs
 * ----------------------------------------------------------------------------------------------------------
 * What GLWrapper::draw does three times after the tracer (GLWrapper.cpp:173-204): bind a program, bind its input
 * textures (all 2-D, GL_LINEAR, CLAMP_TO_EDGE: gen_framebuffer GLWrapper.cpp:209-230 and SMAA_Builder.h:52-83), clear the
 * target to 0 and draw the full-screen quad (GLWrapper.cpp:101-122: position + texture coordinate attributes).
 * Inputs and the target are 8-bit UNORM like the reference's: channels 1 -> GL_R8, 2 -> GL_RG8, 4 -> GL_RGBA8. */
typedef struct glref_pass_tex {
    const char* uniform_name;
    int unit;
    int width, height, channels;   /* 1, 2 or 4 */
    const unsigned char* texels;   /* tightly packed, row 0 = t 0 */
} glref_pass_tex;

int glref_pass(const char* vert_src, const char* frag_src, int w, int h, int n_tex, const glref_pass_tex* tex, int out_channels,
               unsigned char* out)
{
    if (!g_ctx) FAIL("glref_init first");
    char log[3000] = "";
    GLuint vs = compile(GL_VERTEX_SHADER, vert_src, log, sizeof log);
    if (!vs) FAIL("vertex shader: %s", log);
    GLuint fs = compile(GL_FRAGMENT_SHADER, frag_src, log, sizeof log);
    if (!fs) FAIL("fragment shader: %s", log);
    GLuint prog = p_glCreateProgram();
    p_glAttachShader(prog, vs);
    p_glAttachShader(prog, fs);
    p_glLinkProgram(prog);
    GLint ok = 0;
    p_glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) { p_glGetProgramInfoLog(prog, sizeof log, NULL, log); FAIL("link: %s", log); }
    p_glUseProgram(prog);
    if (n_tex > 8) FAIL("too many textures");
    static const GLenum ifmt[5] = {0, GL_R8, GL_RG8, 0, GL_RGBA8}, fmt[5] = {0, GL_RED, GL_RG, 0, GL_RGBA};
    GLuint texid[8] = {0};
    p_glGenTextures(n_tex, texid);
    p_glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    for (int k = 0; k < n_tex; k++) {
        const int c = tex[k].channels;
        if (c != 1 && c != 2 && c != 4) FAIL("pass texture with %d channels", c);
        p_glActiveTexture(GL_TEXTURE0 + tex[k].unit);
        p_glBindTexture(GL_TEXTURE_2D, texid[k]);
        p_glTexImage2D(GL_TEXTURE_2D, 0, (GLint)ifmt[c], tex[k].width, tex[k].height, 0, fmt[c], GL_UNSIGNED_BYTE, tex[k].texels);
        p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
        p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
        p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
        p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
        const GLint loc = p_glGetUniformLocation(prog, tex[k].uniform_name);
        if (loc >= 0) p_glUniform1i(loc, tex[k].unit);
    }
    p_glPixelStorei(GL_UNPACK_ALIGNMENT, 4);
    if (out_channels != 2 && out_channels != 4) FAIL("pass target with %d channels", out_channels);
    GLuint fbo = 0, color = 0, vao = 0, vbo = 0;
    p_glActiveTexture(GL_TEXTURE0 + 15);
    p_glGenTextures(1, &color);
    p_glBindTexture(GL_TEXTURE_2D, color);
    p_glTexImage2D(GL_TEXTURE_2D, 0, (GLint)ifmt[out_channels], w, h, 0, fmt[out_channels], GL_UNSIGNED_BYTE, NULL);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
    p_glGenFramebuffers(1, &fbo);
    p_glBindFramebuffer(GL_FRAMEBUFFER, fbo);
    p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, color, 0);
    if (p_glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) FAIL("framebuffer incomplete");
    p_glBindTexture(GL_TEXTURE_2D, 0);
    static const float quad[] = {-1, -1, 0, 0, 1, -1, 1, 0, 1, 1, 1, 1, -1, -1, 0, 0, 1, 1, 1, 1, -1, 1, 0, 1};
    p_glGenVertexArrays(1, &vao);
    p_glGenBuffers(1, &vbo);
    p_glBindVertexArray(vao);
    p_glBindBuffer(GL_ARRAY_BUFFER, vbo);
    p_glBufferData(GL_ARRAY_BUFFER, sizeof quad, quad, GL_STATIC_DRAW);
    p_glEnableVertexAttribArray(0);
    p_glVertexAttribPointer(0, 2, GL_FLOAT, GL_FALSE, 4 * sizeof(float), (void*)0);
    p_glEnableVertexAttribArray(1);
    p_glVertexAttribPointer(1, 2, GL_FLOAT, GL_FALSE, 4 * sizeof(float), (void*)(2 * sizeof(float)));
    p_glViewport(0, 0, w, h);
    p_glDisable(GL_DEPTH_TEST);
    p_glDisable(GL_BLEND);
    p_glClearColor(0, 0, 0, 0);
    p_glClear(GL_COLOR_BUFFER_BIT);
    p_glDrawArrays(GL_TRIANGLES, 0, 6);
    p_glFinish();
    p_glPixelStorei(GL_PACK_ALIGNMENT, 1);
    p_glReadPixels(0, 0, w, h, fmt[out_channels], GL_UNSIGNED_BYTE, out);
    const GLenum e = p_glGetError();
    p_glBindFramebuffer(GL_FRAMEBUFFER, 0);
    p_glBindVertexArray(0);
    p_glDeleteFramebuffers(1, &fbo);
    p_glDeleteTextures(1, &color);
    p_glDeleteTextures(n_tex, texid);
    p_glDeleteBuffers(1, &vbo);
    p_glUseProgram(0);
    p_glDeleteProgram(prog);
    p_glDeleteShader(vs);
    p_glDeleteShader(fs);
    if (e != GL_NO_ERROR) FAIL("GL error 0x%x", e);
    return 0;
}
