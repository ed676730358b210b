// rtx_capi.cpp -- implementation of the C ABI in include/rtx.h (librtx_hip.so).
//
// Host side of the replaced path: what GLWrapper did with an OpenGL context (reference
// src/GLWrapper.cpp) is done here with a HIP device context -- uniform blocks become one packed
// DevScene blob in HBM (rt_pack.h), textures become RGBA8 arrays, glDrawArrays becomes a kernel
// launch on the context's stream, bracketed by HIP events for the per-draw time.
// There is no CPU fallback: without a usable HIP device rtx_create fails.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is dlopen-ed when a multi-device context asks for it

#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "rtx.h"
#include "rt_kernel.h"
#include "rt_pack.h"
#include "smaa_kernel.h"
#include "bands_kernel.h"

using namespace rtdev;

namespace {

thread_local std::string g_error;
std::mutex g_mutex;
rtx_context* g_current = nullptr;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}
#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) return fail(RTX_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

struct Texture {
    bool cube = false;
    int width = 0, height = 0;  // cube: width = face size
    int wrap = 0;
    int levels = 1;
    int face_mask = 0;
    uint32_t level_off[MAX_MIPS] = {0};
    uint32_t* d_texels = nullptr;
    size_t dwords = 0;
};

const char* const kSamplerNames[] = {"texture_sphere_1", "texture_sphere_2", "texture_sphere_3", "texture_sphere_4", "texture_ring",
                                     "texture_box", "skybox"};
enum { SAMPLER_SKYBOX = 6, SAMPLER_COUNT = 7, UNIT_COUNT = 32, EVENT_RING = 128 };

}  // namespace

struct rtx_context {
    int device = 0;
    int width = 0, height = 0;
    hipStream_t stream = nullptr;
    bool specialized = false;
    rtpack::Defines defines{};
    // uniform blocks: host copies, by binding slot of rt.frag's nine blocks
    std::vector<unsigned char> blocks[rtpack::BLK_COUNT];
    bool block_created[rtpack::BLK_COUNT] = {false};
    bool scene_dirty = true;
    std::vector<unsigned char> blob;
    // device scene: double-buffered pinned staging + device blob
    unsigned char* h_stage[2] = {nullptr, nullptr};
    hipEvent_t stage_done[2] = {nullptr, nullptr};
    size_t stage_cap = 0;
    int stage_next = 0;
    char* d_scene = nullptr;
    size_t d_scene_cap = 0;
    int scene_bytes = 0;
    // cross-stream ordering of the single device scene (rtx_draw_bands may bring its own stream): the stream of the last
    // upload / launch and an event after the last launch; a draw on another stream waits for both (upload_scene, draw_impl)
    hipStream_t upload_stream = nullptr, launch_stream = nullptr;
    int last_stage = -1;
    hipEvent_t launch_done = nullptr;
    // textures
    std::map<uint32_t, Texture> textures;
    uint32_t next_handle = 1;
    int sampler_unit[SAMPLER_COUNT];
    uint32_t unit_texture_2d[UNIT_COUNT];
    uint32_t unit_texture_cube[UNIT_COUNT];
    // colour target
    float* d_fb_f32 = nullptr;
    uint32_t* d_fb_u8 = nullptr;
    // options
    int opt_cull = 1, opt_count = 0, opt_lds = 0, opt_lod = 1, opt_xcd = 0;
    int opt_occ = -1;   // RTX_OPT_HIGH_OCCUPANCY: -1 auto (by primitive count), 0 off, 1 on
    int opt_hot = 1;    // RTX_OPT_HOT_ROWS_FIRST
    unsigned long long* d_counters = nullptr;
    // SMAA post-process (GLWrapper::enable_SMAA + the three passes of GLWrapper.cpp:173-204): smaa_preset < 0 = off
    int smaa_preset = -1;
    bool smaa_tables = false;
    unsigned smaa_frame = 0;
    uint32_t* d_screen = nullptr;    // RGBA8: what the reference shows in its window
    uint16_t* d_edges = nullptr;     // RG8, zero outside the listed pixels
    uint32_t* d_blend = nullptr;     // RGBA8, zero outside the listed pixels
    uint32_t* d_list = nullptr;      // edge pixels of the current frame
    uint32_t* d_smaa_count = nullptr;  // two alternating counters
    uint16_t* d_area = nullptr;
    uint8_t* d_search = nullptr;
    hipEvent_t smaa_start = nullptr, smaa_stop = nullptr;
    bool smaa_timed = false;
    // multi-device (rtx_create_multi): the context the caller holds is rank 0 (the root, which owns the assembled frame); `peers` are
    // the contexts of ranks 1..N-1, ordinary single-device contexts that every scene / texture / option call is forwarded to.
    std::vector<rtx_context*> peers;
    rtx_context* owner = nullptr;      // set on a peer: its root
    int gather_kind = RTX_GATHER_RCCL;
    int band_rows = 0;                 // rows per band of the interleaved split
    void* d_packed[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [format][frame parity] this rank's packed bands, on its own device
    std::vector<void*> d_stage[2][2];  // root only: [format][frame parity][rank] landing buffers for the peers' bands, on the root's device
    ncclComm_t comm = nullptr;
    hipStream_t gather_stream = nullptr;   // root only: receives + band placement run here, beside the next frame's trace
    hipEvent_t traced[2] = {nullptr, nullptr}, gathered[2] = {nullptr, nullptr};
    unsigned frame_no = 0;
    float last_gather_ms = 0.0f;
    hipEvent_t gather_start = nullptr, gather_stop = nullptr;
    bool gather_timed = false;
    // timing
    hipEvent_t ev_start[EVENT_RING], ev_stop[EVENT_RING];
    int ev_head = 0, ev_pending = 0;
    float last_ms = 0.0f;
    uint32_t launches = 0;
};

namespace {

int find_block(const char* name)
{
    for (int b = 0; b < rtpack::BLK_COUNT; b++)
        if (name && !std::strcmp(name, rtpack::kBlockNames[b])) return b;
    return -1;
}
int find_sampler(const char* name)
{
    for (int s = 0; s < SAMPLER_COUNT; s++)
        if (name && !std::strcmp(name, kSamplerNames[s])) return s;
    return -1;
}

int use_device(rtx_context* ctx)
{
    HIP_TRY(hipSetDevice(ctx->device));
    return RTX_OK;
}

int upload_scene(rtx_context* ctx, hipStream_t stream)
{
    for (int b = 0; b < rtpack::BLK_COUNT; b++)
        if (!ctx->block_created[b]) return fail(RTX_ERR_ORDER, "draw before init_buffer(\"%s\")", rtpack::kBlockNames[b]);
    std::string err;
    if (!rtpack::pack_scene(ctx->defines, ctx->blocks, ctx->blob, err)) return fail(RTX_ERR_INVALID, "%s", err.c_str());
    const size_t n = ctx->blob.size();
    if (n > ctx->stage_cap) {
        for (int k = 0; k < 2; k++) {
            if (ctx->stage_done[k]) HIP_TRY(hipEventSynchronize(ctx->stage_done[k]));
            if (ctx->h_stage[k]) HIP_TRY(hipHostFree(ctx->h_stage[k]));
            ctx->h_stage[k] = nullptr;
        }
        const size_t cap = (n + 4095) & ~static_cast<size_t>(4095);
        for (int k = 0; k < 2; k++) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_stage[k]), cap, hipHostMallocDefault));
        ctx->stage_cap = cap;
    }
    if (n > ctx->d_scene_cap) {
        HIP_TRY(hipDeviceSynchronize());
        if (ctx->d_scene) HIP_TRY(hipFree(ctx->d_scene));
        const size_t cap = (n + 4095) & ~static_cast<size_t>(4095);
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_scene), cap));
        ctx->d_scene_cap = cap;
    }
    // the device scene is about to be overwritten: a kernel still reading it on ANOTHER stream has to finish first
    if (ctx->launch_stream && ctx->launch_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, ctx->launch_done, 0));
    const int k = ctx->stage_next;
    ctx->stage_next ^= 1;
    HIP_TRY(hipEventSynchronize(ctx->stage_done[k]));  // staging buffer k is free again
    std::memcpy(ctx->h_stage[k], ctx->blob.data(), n);
    HIP_TRY(hipMemcpyAsync(ctx->d_scene, ctx->h_stage[k], n, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(ctx->stage_done[k], stream));
    ctx->upload_stream = stream;
    ctx->last_stage = k;
    ctx->scene_bytes = static_cast<int>(n);
    ctx->scene_dirty = false;
    return RTX_OK;
}

void fill_tex_table(rtx_context* ctx, TexTable& T)
{
    std::memset(&T, 0, sizeof T);
    for (int s = 0; s < TEX_SLOTS; s++) {
        const int unit = ctx->sampler_unit[s];
        if (unit < 0 || unit >= UNIT_COUNT) continue;
        auto it = ctx->textures.find(ctx->unit_texture_2d[unit]);
        if (it == ctx->textures.end() || it->second.cube) continue;
        const Texture& t = it->second;
        DevTexture& d = T.tex[s];
        d.texels = t.d_texels;
        d.width = t.width;
        d.height = t.height;
        d.wrap = t.wrap;
        d.fwidth = static_cast<float>(t.width);
        d.fheight = static_cast<float>(t.height);
        d.levels = ctx->opt_lod ? t.levels : 1;
        std::memcpy(d.level_off, t.level_off, sizeof d.level_off);
    }
    T.lod = ctx->opt_lod;
    const int unit = ctx->sampler_unit[SAMPLER_SKYBOX];
    if (unit >= 0 && unit < UNIT_COUNT) {
        auto it = ctx->textures.find(ctx->unit_texture_cube[unit]);
        if (it != ctx->textures.end() && it->second.cube) {
            T.sky.texels = it->second.d_texels;
            T.sky.size = it->second.width;
            T.sky.fsize = static_cast<float>(it->second.width);
            T.sky.face_mask = it->second.face_mask;
        }
    }
}

int drain_events(rtx_context* ctx)
{
    while (ctx->ev_pending > 0) {
        const int idx = (ctx->ev_head - ctx->ev_pending + EVENT_RING * 2) % EVENT_RING;
        HIP_TRY(hipEventSynchronize(ctx->ev_stop[idx]));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_start[idx], ctx->ev_stop[idx]));
        ctx->last_ms = ms;
        ctx->ev_pending--;
    }
    return RTX_OK;
}

// Which workgroup rows of this launch show a torus (the long-running tiles, see rt_kernel.hip): the framebuffer rows covered
// by the tori's bounding spheres seen from the camera, mapped through the band layout. A scheduling hint only -- computed in
// double precision from the packed scene, padded, and ignored when a torus reaches behind the camera.
void hot_rows(rtx_context* ctx, RtLaunchParams& p)
{
    p.hot_row0 = p.hot_rows = 0;
    const DevSceneHeader* h = reinterpret_cast<const DevSceneHeader*>(ctx->blob.data());
    if (ctx->blob.size() < sizeof(DevSceneHeader) || h->n_torus <= 0) return;
    const double qx = h->cam_quat.x, qy = h->cam_quat.y, qz = h->cam_quat.z, qw = h->cam_quat.w;
    const double H = ctx->height;
    double ylo = 1e30, yhi = -1e30;
    const DevTorus* tori = reinterpret_cast<const DevTorus*>(ctx->blob.data() + h->off_torus);
    for (int i = 0; i < h->n_torus; i++) {
        // camera space: the shader rotates the view vector by q, so a world offset goes back with the conjugate
        const double vx = tori[i].pos.x - h->cam_pos.x, vy = tori[i].pos.y - h->cam_pos.y, vz = tori[i].pos.z - h->cam_pos.z;
        const double tx = qw * vx - (qy * vz - qz * vy), ty = qw * vy - (qz * vx - qx * vz), tz = qw * vz - (qx * vy - qy * vx), tw = qx * vx + qy * vy + qz * vz;
        const double cy = tw * qy + ty * qw + (tz * qx - tx * qz), cz = tw * qz + tz * qw + (tx * qy - ty * qx);   // (conj(q) v) q
        const double rb = std::fabs(static_cast<double>(tori[i].radii.x)) + std::fabs(static_cast<double>(tori[i].radii.y));
        if (!(cz - rb > 1e-3) || !(rb < 1e30)) return;   // reaches the camera plane: no useful extent
        const double a = (cy - rb) / (cz - rb), b = (cy - rb) / (cz + rb), c = (cy + rb) / (cz - rb), d = (cy + rb) / (cz + rb);
        ylo = std::fmin(ylo, std::fmin(std::fmin(a, b), std::fmin(c, d)));
        yhi = std::fmax(yhi, std::fmax(std::fmax(a, b), std::fmax(c, d)));
    }
    const double r0 = ylo * H + 0.5 * H - 4.0, r1 = yhi * H + 0.5 * H + 4.0;   // framebuffer rows, padded
    if (!(r1 > 0.0) || !(r0 < H)) return;
    const int fy0 = r0 < 0.0 ? 0 : static_cast<int>(r0), fy1 = r1 > H ? ctx->height : static_cast<int>(r1) + 1;
    const int grid_y = (p.rows_local + 7) / 8;
    int first = -1, last = -1;
    for (int by = 0; by < grid_y; by++) {   // same mapping as the kernel: local workgroup row -> first framebuffer row
        const int band_j = (by * 8) / p.band_rows;
        const int y0 = (p.band_first + band_j * p.band_stride) * p.band_rows + (by * 8 - band_j * p.band_rows);
        if (y0 + 8 > fy0 && y0 < fy1) { if (first < 0) first = by; last = by; }
    }
    if (first < 0 || (last - first + 1) * 2 > grid_y) return;   // nothing, or "hot" is most of the launch: order is moot
    p.hot_row0 = first;
    p.hot_rows = last - first + 1;
}

int draw_impl(rtx_context* ctx, int band_rows, int band_first, int band_stride, float* out_f32, uint32_t* out_u8, hipStream_t stream)
{
    if (!ctx->specialized) return fail(RTX_ERR_ORDER, "draw before init_shaders/rtx_specialize");
    int st = use_device(ctx);
    if (st) return st;
    if (band_rows <= 0 || (band_rows % 8) != 0 || band_first < 0 || band_stride <= 0) return fail(RTX_ERR_INVALID, "bad band arguments");
    if (ctx->scene_dirty) {
        st = upload_scene(ctx, stream);  // same stream as the launch: ordered before it, and after earlier draws on it
        if (st) return st;
    }
    else if (ctx->last_stage >= 0 && ctx->upload_stream != stream) {
        HIP_TRY(hipStreamWaitEvent(stream, ctx->stage_done[ctx->last_stage], 0));  // the scene was uploaded on another stream
    }
    const int n_bands_total = (ctx->height + band_rows - 1) / band_rows;
    int rows_local = 0;
    for (int b = band_first; b < n_bands_total; b += band_stride) {
        const int y0 = b * band_rows;
        const int y1 = y0 + band_rows < ctx->height ? y0 + band_rows : ctx->height;
        rows_local += y1 - y0;
    }
    RtLaunchParams p;
    std::memset(&p, 0, sizeof p);
    p.scene = ctx->d_scene;
    p.scene_bytes = ctx->scene_bytes;
    p.fb_w = ctx->width;
    p.fb_h = ctx->height;
    p.band_rows = band_rows;
    p.band_first = band_first;
    p.band_stride = band_stride;
    p.rows_local = rows_local;
    p.xcd_remap = ctx->opt_xcd;
    if (ctx->opt_hot && !ctx->opt_xcd && band_stride >= 4) hot_rows(ctx, p);   // small launches only, see rt_kernel.hip
    p.out_f32 = out_f32;
    p.out_u8 = out_u8;
    p.counters = ctx->d_counters;
    fill_tex_table(ctx, p.tex);
    if (ctx->opt_count) HIP_TRY(hipMemsetAsync(ctx->d_counters, 0, 4 * sizeof(unsigned long long), stream));
    if (ctx->ev_pending == EVENT_RING) {  // ring full: retire the oldest pair only (recorded EVENT_RING launches ago, long finished)
        const int idx = (ctx->ev_head - ctx->ev_pending + EVENT_RING * 2) % EVENT_RING;
        HIP_TRY(hipEventSynchronize(ctx->ev_stop[idx]));
        ctx->ev_pending--;
    }
    const int e = ctx->ev_head;
    HIP_TRY(hipEventRecord(ctx->ev_start[e], stream));
    // long primitive tables (quadric-/torus-heavy scenes): the 7-waves-per-SIMD build of the kernel hides the table walks
    const rtpack::Defines& df = ctx->defines;
    const int n_prims = df.sphere_size + df.plane_size + df.surface_size + df.box_size + df.torus_size + df.ring_size;
    const bool high_occ = ctx->opt_occ < 0 ? n_prims >= 32 : ctx->opt_occ != 0;
    HIP_TRY(rt_launch_trace(p, ctx->opt_cull != 0, ctx->opt_count != 0, ctx->opt_lds != 0, high_occ, stream));
    HIP_TRY(hipEventRecord(ctx->ev_stop[e], stream));
    HIP_TRY(hipEventRecord(ctx->launch_done, stream));
    ctx->launch_stream = stream;
    ctx->ev_head = (ctx->ev_head + 1) % EVENT_RING;
    ctx->ev_pending++;
    ctx->launches++;
    return RTX_OK;
}

int smaa_alloc(rtx_context* ctx)
{
    if (ctx->d_screen) return RTX_OK;
    const size_t px = static_cast<size_t>(ctx->width) * ctx->height;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_screen), px * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_edges), px * 2));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_blend), px * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_list), smaa_segment_capacity(ctx->width, ctx->height) * SMAA_SEGMENTS * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_smaa_count), 2 * SMAA_SEGMENTS * sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(ctx->d_edges, 0, px * 2, ctx->stream));      // the sparse passes keep both textures zero outside the
    HIP_TRY(hipMemsetAsync(ctx->d_blend, 0, px * 4, ctx->stream));      // current frame's edge pixels (smaa_kernel.hip)
    HIP_TRY(hipMemsetAsync(ctx->d_smaa_count, 0, 2 * SMAA_SEGMENTS * sizeof(uint32_t), ctx->stream));
    HIP_TRY(hipEventCreate(&ctx->smaa_start));
    HIP_TRY(hipEventCreate(&ctx->smaa_stop));
    ctx->smaa_frame = 0;
    return RTX_OK;
}

// The three passes after the tracer (GLWrapper.cpp:173-204) on the context's RGBA8 colour target, into the screen buffer.
int smaa_resolve(rtx_context* ctx, hipStream_t stream)
{
    if (!ctx->smaa_tables) return fail(RTX_ERR_ORDER, "SMAA is enabled but rtx_smaa_set_tables has not supplied the area / search tables");
    int st = smaa_alloc(ctx);
    if (st) return st;
    if (stream != ctx->stream) return fail(RTX_ERR_INVALID, "the SMAA resolve runs on the context's own stream");
    SmaaBuffers b;
    b.w = ctx->width;
    b.h = ctx->height;
    b.color = ctx->d_fb_u8;
    b.screen = ctx->d_screen;
    b.edges = ctx->d_edges;
    b.blend = ctx->d_blend;
    b.list = ctx->d_list;
    b.segment_capacity = smaa_segment_capacity(ctx->width, ctx->height);
    b.count = ctx->d_smaa_count;
    b.area = ctx->d_area;
    b.search = ctx->d_search;
    HIP_TRY(hipEventRecord(ctx->smaa_start, stream));
    HIP_TRY(smaa_launch(b, ctx->smaa_preset, ctx->smaa_frame, stream));
    HIP_TRY(hipEventRecord(ctx->smaa_stop, stream));
    ctx->smaa_frame++;
    ctx->smaa_timed = true;
    return RTX_OK;
}

}  // namespace

extern "C" {

const char* rtx_last_error(void) { return g_error.c_str(); }
const char* rtx_version(void) { return "rtx-hip 0.1 (gfx950, HIP tracer for the rt.frag path)"; }

int rtx_create(int width, int height, int device, rtx_context** out)
{
    if (!out || width <= 0 || height <= 0) return fail(RTX_ERR_INVALID, "rtx_create: bad arguments");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(RTX_ERR_DEVICE, "no HIP device available (%s); this library has no CPU fallback", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(RTX_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
    rtx_context* ctx = new rtx_context();
    ctx->device = device;
    ctx->width = width;
    ctx->height = height;
    for (int s = 0; s < SAMPLER_COUNT; s++) ctx->sampler_unit[s] = 0;  // GLSL samplers default to unit 0
    std::memset(ctx->unit_texture_2d, 0, sizeof ctx->unit_texture_2d);
    std::memset(ctx->unit_texture_cube, 0, sizeof ctx->unit_texture_cube);
    auto bail = [&](hipError_t err, const char* what) {
        fail(RTX_ERR_DEVICE, "%s failed: %s", what, hipGetErrorString(err));
        rtx_destroy(ctx);
        return RTX_ERR_DEVICE;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    const size_t px = static_cast<size_t>(width) * height;
    if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->d_fb_f32), px * 16)) != hipSuccess) return bail(e, "hipMalloc(framebuffer f32)");
    if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->d_fb_u8), px * 4)) != hipSuccess) return bail(e, "hipMalloc(framebuffer u8)");
    if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->d_counters), 4 * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc(counters)");
    if ((e = hipMemset(ctx->d_counters, 0, 4 * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMemset");
    for (int k = 0; k < 2; k++)
        if ((e = hipEventCreateWithFlags(&ctx->stage_done[k], hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&ctx->launch_done, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    for (int k = 0; k < EVENT_RING; k++) {
        if ((e = hipEventCreate(&ctx->ev_start[k])) != hipSuccess) return bail(e, "hipEventCreate");
        if ((e = hipEventCreate(&ctx->ev_stop[k])) != hipSuccess) return bail(e, "hipEventCreate");
    }
    {
        std::lock_guard<std::mutex> lk(g_mutex);
        g_current = ctx;
    }
    *out = ctx;
    return RTX_OK;
}

void rtx_destroy(rtx_context* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->textures)
        if (kv.second.d_texels) (void)hipFree(kv.second.d_texels);
    if (ctx->d_scene) (void)hipFree(ctx->d_scene);
    for (int k = 0; k < 2; k++) {
        if (ctx->h_stage[k]) (void)hipHostFree(ctx->h_stage[k]);
        if (ctx->stage_done[k]) (void)hipEventDestroy(ctx->stage_done[k]);
    }
    if (ctx->launch_done) (void)hipEventDestroy(ctx->launch_done);
    if (ctx->d_fb_f32) (void)hipFree(ctx->d_fb_f32);
    if (ctx->d_fb_u8) (void)hipFree(ctx->d_fb_u8);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    for (void* p : {static_cast<void*>(ctx->d_screen), static_cast<void*>(ctx->d_edges), static_cast<void*>(ctx->d_blend), static_cast<void*>(ctx->d_list),
                    static_cast<void*>(ctx->d_smaa_count), static_cast<void*>(ctx->d_area), static_cast<void*>(ctx->d_search)})
        if (p) (void)hipFree(p);
    if (ctx->smaa_start) (void)hipEventDestroy(ctx->smaa_start);
    if (ctx->smaa_stop) (void)hipEventDestroy(ctx->smaa_stop);
    for (int k = 0; k < EVENT_RING; k++) {
        if (ctx->ev_start[k]) (void)hipEventDestroy(ctx->ev_start[k]);
        if (ctx->ev_stop[k]) (void)hipEventDestroy(ctx->ev_stop[k]);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    {
        std::lock_guard<std::mutex> lk(g_mutex);
        if (g_current == ctx) g_current = nullptr;
    }
    delete ctx;
}

rtx_context* rtx_current(void)
{
    std::lock_guard<std::mutex> lk(g_mutex);
    return g_current;
}
int rtx_make_current(rtx_context* ctx)
{
    std::lock_guard<std::mutex> lk(g_mutex);
    g_current = ctx;
    return RTX_OK;
}
int rtx_get_size(rtx_context* ctx, int* width, int* height)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "null context");
    if (width) *width = ctx->width;
    if (height) *height = ctx->height;
    return RTX_OK;
}

int rtx_specialize(rtx_context* ctx, const rtx_defines* d)
{
    if (!ctx || !d) return fail(RTX_ERR_INVALID, "rtx_specialize: null argument");
    static_assert(sizeof(rtx_defines) == sizeof(rtpack::Defines), "defines layout");
    const int32_t* c = &d->sphere_size;
    for (int k = 0; k < 9; k++)
        if (c[k] < 0 || c[k] > (1 << 20)) return fail(RTX_ERR_INVALID, "rtx_specialize: count %d out of range", k);
    // The trace loop is bounded by RT_SEGMENT_CAP main-loop trips per pixel (refraction does i--, so the shader's own loop has
    // no bound: trap T2) and keeps i in 16 bits: a bounce depth beyond the cap could not be honoured and is refused rather
    // than silently truncated. (The reference's default is 5, SceneManager.cpp:233.)
    if (d->iterations > RTX_MAX_ITERATIONS) return fail(RTX_ERR_INVALID, "rtx_specialize: %d iterations exceed the supported maximum of %d", d->iterations, RTX_MAX_ITERATIONS);
    static_assert(RTX_MAX_ITERATIONS == RT_SEGMENT_CAP, "rtx.h documents the kernel's segment cap");
    std::memcpy(&ctx->defines, d, sizeof *d);
    ctx->specialized = true;
    ctx->scene_dirty = true;
    return RTX_OK;
}

int rtx_block_create(rtx_context* ctx, const char* name, int /*binding_point*/, size_t size, const void* data, uint32_t* handle)
{
    if (!ctx || !handle) return fail(RTX_ERR_INVALID, "rtx_block_create: null argument");
    if (!ctx->specialized) return fail(RTX_ERR_ORDER, "init_buffer before init_shaders (block names are looked up in the program)");
    const int b = find_block(name);
    if (b < 0) return fail(RTX_ERR_NAME, "Invalid ubo block name '%s'", name ? name : "(null)");
    ctx->blocks[b].assign(size, 0);
    if (data && size) std::memcpy(ctx->blocks[b].data(), data, size);
    ctx->block_created[b] = true;
    ctx->scene_dirty = true;
    *handle = static_cast<uint32_t>(b + 1);  // handles 1..9 (0 is GL's "no buffer")
    return RTX_OK;
}

int rtx_block_update(rtx_context* ctx, uint32_t handle, size_t size, const void* data)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "rtx_block_update: no current context");
    if (handle < 1 || handle > rtpack::BLK_COUNT || !ctx->block_created[handle - 1]) return fail(RTX_ERR_HANDLE, "unknown block handle %u", handle);
    std::vector<unsigned char>& blk = ctx->blocks[handle - 1];
    if (size > blk.size()) return fail(RTX_ERR_INVALID, "update of %zu bytes exceeds the block's %zu bytes (glBufferSubData would raise GL_INVALID_VALUE)", size, blk.size());
    if (size && !data) return fail(RTX_ERR_INVALID, "null data");
    if (size) std::memcpy(blk.data(), data, size);
    ctx->scene_dirty = true;
    return RTX_OK;
}

int rtx_texture2d_create(rtx_context* ctx, int width, int height, int channels, const uint8_t* texels, int wrap, uint32_t* handle)
{
    if (!ctx || !handle || !texels) return fail(RTX_ERR_INVALID, "rtx_texture2d_create: null argument");
    if (width <= 0 || height <= 0 || (channels != 1 && channels != 3 && channels != 4)) return fail(RTX_ERR_INVALID, "unsupported texture %dx%d, %d channels", width, height, channels);
    int st = use_device(ctx);
    if (st) return st;
    Texture t;
    t.width = width;
    t.height = height;
    t.wrap = wrap == RTX_WRAP_CLAMP_TO_EDGE ? 1 : 0;
    std::vector<uint32_t> host(static_cast<size_t>(width) * height);
    rtpack::to_rgba8(texels, width, height, channels, host.data());
    t.levels = rtpack::build_mip_chain(host, width, height, t.level_off, MAX_MIPS);  // glGenerateMipmap (GLWrapper.cpp:337)
    t.dwords = host.size();
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t.d_texels), t.dwords * 4));
    HIP_TRY(hipMemcpy(t.d_texels, host.data(), t.dwords * 4, hipMemcpyHostToDevice));
    const uint32_t h = ctx->next_handle++;
    ctx->textures[h] = t;
    *handle = h;
    return RTX_OK;
}

int rtx_cubemap_create(rtx_context* ctx, int face_size, int channels, const uint8_t* const faces[6], int /*gen_mipmap*/, uint32_t* handle)
{
    if (!ctx || !handle || !faces) return fail(RTX_ERR_INVALID, "rtx_cubemap_create: null argument");
    int st = use_device(ctx);
    if (st) return st;
    Texture t;
    t.cube = true;
    t.wrap = 1;
    bool any = false;
    for (int f = 0; f < 6; f++) any = any || faces[f];
    if (any && (face_size <= 0 || (channels != 1 && channels != 3 && channels != 4))) return fail(RTX_ERR_INVALID, "unsupported cubemap face %d, %d channels", face_size, channels);
    if (any) {
        t.width = t.height = face_size;
        const size_t fsz = static_cast<size_t>(face_size) * face_size;
        std::vector<uint32_t> host(fsz * 6, 0u);
        for (int f = 0; f < 6; f++)
            if (faces[f]) { rtpack::to_rgba8(faces[f], face_size, face_size, channels, host.data() + fsz * f); t.face_mask |= 1 << f; }
        t.dwords = host.size();
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t.d_texels), t.dwords * 4));
        HIP_TRY(hipMemcpy(t.d_texels, host.data(), t.dwords * 4, hipMemcpyHostToDevice));
    }
    const uint32_t h = ctx->next_handle++;
    ctx->textures[h] = t;
    *handle = h;
    return RTX_OK;
}

int rtx_sampler_unit(rtx_context* ctx, const char* sampler_name, int unit)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "null context");
    if (!ctx->specialized) return fail(RTX_ERR_ORDER, "sampler uniform set before init_shaders");
    const int s = find_sampler(sampler_name);
    if (s < 0) return fail(RTX_ERR_NAME, "unknown sampler '%s'", sampler_name ? sampler_name : "(null)");
    if (unit < 0 || unit >= UNIT_COUNT) return fail(RTX_ERR_INVALID, "texture unit %d out of range", unit);
    ctx->sampler_unit[s] = unit;
    return RTX_OK;
}

int rtx_bind_texture(rtx_context* ctx, int unit, uint32_t handle)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "rtx_bind_texture: no current context");
    if (unit < 0 || unit >= UNIT_COUNT) return fail(RTX_ERR_INVALID, "texture unit %d out of range", unit);
    if (handle == 0) { ctx->unit_texture_2d[unit] = 0; ctx->unit_texture_cube[unit] = 0; return RTX_OK; }  // glBindTexture(target, 0): there is no target argument here, so both bindings of the unit are cleared
    auto it = ctx->textures.find(handle);
    if (it == ctx->textures.end()) return fail(RTX_ERR_HANDLE, "unknown texture handle %u", handle);
    if (it->second.cube) ctx->unit_texture_cube[unit] = handle;
    else ctx->unit_texture_2d[unit] = handle;
    return RTX_OK;
}

int rtx_texture_destroy(rtx_context* ctx, uint32_t handle)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "null context");
    auto it = ctx->textures.find(handle);
    if (it == ctx->textures.end()) return fail(RTX_ERR_HANDLE, "unknown texture handle %u", handle);
    int st = use_device(ctx);
    if (st) return st;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (it->second.d_texels) HIP_TRY(hipFree(it->second.d_texels));
    for (int u = 0; u < UNIT_COUNT; u++) {
        if (ctx->unit_texture_2d[u] == handle) ctx->unit_texture_2d[u] = 0;
        if (ctx->unit_texture_cube[u] == handle) ctx->unit_texture_cube[u] = 0;
    }
    ctx->textures.erase(it);
    return RTX_OK;
}

int rtx_set_option(rtx_context* ctx, int option, int value)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "null context");
    switch (option) {
        case RTX_OPT_CULL: ctx->opt_cull = value != 0; break;
        case RTX_OPT_COUNT_RAYS: ctx->opt_count = value != 0; break;
        case RTX_OPT_SCENE_LDS: ctx->opt_lds = value != 0; break;
        case RTX_OPT_TEXTURE_LOD: ctx->opt_lod = value != 0; break;
        case RTX_OPT_XCD_REMAP: ctx->opt_xcd = value != 0; break;
        case RTX_OPT_HIGH_OCCUPANCY: ctx->opt_occ = value < 0 ? -1 : (value != 0); break;
        case RTX_OPT_HOT_ROWS_FIRST: ctx->opt_hot = value != 0; break;
        default: return fail(RTX_ERR_INVALID, "unknown option %d", option);
    }
    return RTX_OK;
}
int rtx_get_option(rtx_context* ctx, int option, int* value)
{
    if (!ctx || !value) return fail(RTX_ERR_INVALID, "null argument");
    switch (option) {
        case RTX_OPT_CULL: *value = ctx->opt_cull; break;
        case RTX_OPT_COUNT_RAYS: *value = ctx->opt_count; break;
        case RTX_OPT_SCENE_LDS: *value = ctx->opt_lds; break;
        case RTX_OPT_TEXTURE_LOD: *value = ctx->opt_lod; break;
        case RTX_OPT_XCD_REMAP: *value = ctx->opt_xcd; break;
        case RTX_OPT_HIGH_OCCUPANCY: *value = ctx->opt_occ; break;
        case RTX_OPT_HOT_ROWS_FIRST: *value = ctx->opt_hot; break;
        default: return fail(RTX_ERR_INVALID, "unknown option %d", option);
    }
    return RTX_OK;
}

int rtx_draw(rtx_context* ctx)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "null context");
    const int band = ((ctx->height + 7) / 8) * 8;
    int st = draw_impl(ctx, band, 0, 1, ctx->d_fb_f32, ctx->d_fb_u8, ctx->stream);
    if (st == RTX_OK && ctx->smaa_preset >= 0) st = smaa_resolve(ctx, ctx->stream);   // GLWrapper.cpp:168-204: the passes follow the tracer
    return st;
}

/* ---- SMAA ---- */
int rtx_smaa_set_tables(rtx_context* ctx, const uint8_t* area_rg8, int area_w, int area_h, const uint8_t* search_r8, int search_w, int search_h)
{
    if (!ctx || !area_rg8 || !search_r8) return fail(RTX_ERR_INVALID, "rtx_smaa_set_tables: null argument");
    if (area_w != 160 || area_h != 560 || search_w != 64 || search_h != 16)
        return fail(RTX_ERR_INVALID, "SMAA tables must be 160x560 (RG8) and 64x16 (R8), the sizes SMAA.h addresses (SMAA.h:519-522); got %dx%d and %dx%d",
                    area_w, area_h, search_w, search_h);
    int st = use_device(ctx);
    if (st) return st;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (!ctx->d_area) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_area), 160 * 560 * 2));
    if (!ctx->d_search) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_search), 64 * 16));
    HIP_TRY(hipMemcpy(ctx->d_area, area_rg8, 160 * 560 * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_search, search_r8, 64 * 16, hipMemcpyHostToDevice));
    ctx->smaa_tables = true;
    return RTX_OK;
}

int rtx_enable_smaa(rtx_context* ctx, int preset)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "null context");
    if (preset < -1 || preset > RTX_SMAA_ULTRA) return fail(RTX_ERR_INVALID, "unknown SMAA preset %d", preset);
    ctx->smaa_preset = preset;
    if (preset >= 0) {
        int st = use_device(ctx);
        if (st) return st;
        return smaa_alloc(ctx);
    }
    return RTX_OK;
}

int rtx_smaa_resolve(rtx_context* ctx)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "null context");
    if (ctx->smaa_preset < 0) return fail(RTX_ERR_ORDER, "rtx_smaa_resolve: SMAA is not enabled (rtx_enable_smaa)");
    int st = use_device(ctx);
    if (st) return st;
    return smaa_resolve(ctx, ctx->stream);
}

int rtx_write_pixels(rtx_context* ctx, int format, const void* src_host, size_t src_bytes)
{
    if (!ctx || !src_host) return fail(RTX_ERR_INVALID, "rtx_write_pixels: null argument");
    if (format != RTX_RGBA8) return fail(RTX_ERR_INVALID, "rtx_write_pixels: only the RGBA8 colour target can be written");
    const size_t need = static_cast<size_t>(ctx->width) * ctx->height * 4;
    if (src_bytes < need) return fail(RTX_ERR_INVALID, "source holds %zu bytes, %zu needed", src_bytes, need);
    int st = use_device(ctx);
    if (st) return st;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(ctx->d_fb_u8, src_host, need, hipMemcpyHostToDevice));
    return RTX_OK;
}

int rtx_draw_bands(rtx_context* ctx, int band_rows, int band_first, int band_stride, void* dst_device, int format, void* stream)
{
    if (!ctx || !dst_device) return fail(RTX_ERR_INVALID, "rtx_draw_bands: null argument");
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    if (format == RTX_RGBA32F) return draw_impl(ctx, band_rows, band_first, band_stride, static_cast<float*>(dst_device), nullptr, s);
    if (format == RTX_RGBA8) return draw_impl(ctx, band_rows, band_first, band_stride, nullptr, static_cast<uint32_t*>(dst_device), s);
    return fail(RTX_ERR_INVALID, "unknown format %d", format);
}

int rtx_finish(rtx_context* ctx)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "null context");
    int st = use_device(ctx);
    if (st) return st;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RTX_OK;
}

int rtx_read_pixels(rtx_context* ctx, int format, void* dst_host, size_t dst_bytes)
{
    if (!ctx || !dst_host) return fail(RTX_ERR_INVALID, "rtx_read_pixels: null argument");
    int st = use_device(ctx);
    if (st) return st;
    const size_t px = static_cast<size_t>(ctx->width) * ctx->height;
    const void* src = nullptr;
    size_t need = 0;
    switch (format) {
        case RTX_RGBA32F: src = ctx->d_fb_f32; need = px * 16; break;
        case RTX_RGBA8: src = ctx->d_fb_u8; need = px * 4; break;
        case RTX_SCREEN_RGBA8: src = (ctx->smaa_preset >= 0 && ctx->d_screen) ? ctx->d_screen : ctx->d_fb_u8; need = px * 4; break;
        case RTX_SMAA_EDGES_RG8: src = ctx->d_edges; need = px * 2; break;
        case RTX_SMAA_WEIGHTS_RGBA8: src = ctx->d_blend; need = px * 4; break;
        default: return fail(RTX_ERR_INVALID, "unknown format %d", format);
    }
    if (!src) return fail(RTX_ERR_ORDER, "format %d needs SMAA to be enabled (rtx_enable_smaa)", format);
    if (dst_bytes < need) return fail(RTX_ERR_INVALID, "destination holds %zu bytes, %zu needed", dst_bytes, need);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(dst_host, src, need, hipMemcpyDeviceToHost));
    return RTX_OK;
}

int rtx_framebuffer_device(rtx_context* ctx, int format, void** device_ptr)
{
    if (!ctx || !device_ptr) return fail(RTX_ERR_INVALID, "null argument");
    if (format == RTX_RGBA32F) *device_ptr = ctx->d_fb_f32;
    else if (format == RTX_RGBA8) *device_ptr = ctx->d_fb_u8;
    else if (format == RTX_SCREEN_RGBA8) *device_ptr = (ctx->smaa_preset >= 0 && ctx->d_screen) ? ctx->d_screen : ctx->d_fb_u8;
    else return fail(RTX_ERR_INVALID, "unknown format %d", format);
    return RTX_OK;
}

int rtx_get_stats(rtx_context* ctx, rtx_stats* out)
{
    if (!ctx || !out) return fail(RTX_ERR_INVALID, "null argument");
    int st = use_device(ctx);
    if (st) return st;
    st = drain_events(ctx);
    if (st) return st;
    std::memset(out, 0, sizeof *out);
    out->last_draw_ms = ctx->last_ms;
    out->launches = ctx->launches;
    if (ctx->smaa_timed) {
        HIP_TRY(hipEventSynchronize(ctx->smaa_stop));
        HIP_TRY(hipEventElapsedTime(&out->last_smaa_ms, ctx->smaa_start, ctx->smaa_stop));
        uint32_t n[SMAA_SEGMENTS];
        HIP_TRY(hipMemcpy(n, ctx->d_smaa_count + ((ctx->smaa_frame - 1u) & 1u) * SMAA_SEGMENTS, sizeof n, hipMemcpyDeviceToHost));
        for (int k = 0; k < SMAA_SEGMENTS; k++) out->smaa_edge_pixels += n[k];
    }
    if (ctx->opt_count) {
        unsigned long long c[4];
        HIP_TRY(hipMemcpy(c, ctx->d_counters, sizeof c, hipMemcpyDeviceToHost));
        out->rays_closest = c[0];
        out->rays_shadow = c[1];
        out->rays_shadow_cast = c[2];
        out->torus_solves = c[3];
    }
    return RTX_OK;
}

// Extra diagnostics (not part of the reference surface) ---------------------------------------
// Sum of the HIP-event durations of the `n` most recent draws (n <= 128), for benches that time
// K draws back to back. Returns RTX_ERR_INVALID if fewer than n draws are pending.
RTX_API int rtx_sum_recent_draw_ms(rtx_context* ctx, int n, float* sum_ms)
{
    if (!ctx || !sum_ms || n <= 0 || n > ctx->ev_pending) return fail(RTX_ERR_INVALID, "rtx_sum_recent_draw_ms: %d draws requested, %d pending", n, ctx ? ctx->ev_pending : 0);
    int st = use_device(ctx);
    if (st) return st;
    float total = 0.0f;
    for (int k = 0; k < n; k++) {
        const int idx = (ctx->ev_head - 1 - k + EVENT_RING * 2) % EVENT_RING;
        HIP_TRY(hipEventSynchronize(ctx->ev_stop[idx]));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_start[idx], ctx->ev_stop[idx]));
        total += ms;
        if (k == 0) ctx->last_ms = ms;
    }
    ctx->ev_pending = 0;
    *sum_ms = total;
    return RTX_OK;
}

// Runs the device-side exhaustive check of the divide-free byte->float conversion.
RTX_API int rtx_selftest(rtx_context* ctx, int* mismatches)
{
    if (!ctx || !mismatches) return fail(RTX_ERR_INVALID, "null argument");
    int st = use_device(ctx);
    if (st) return st;
    int* d = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), sizeof(int)));
    HIP_TRY(hipMemsetAsync(d, 0, sizeof(int), ctx->stream));
    HIP_TRY(rt_launch_selftest(d, ctx->stream));
    HIP_TRY(hipMemcpyAsync(mismatches, d, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(d));
    return RTX_OK;
}

}  // extern "C"
