// rtx_capi.cpp -- implementation of the C ABI in include/rtx.h (librtx_hip.so).
//
// Host side of the replaced path: what GLWrapper did with an OpenGL context (reference
// src/GLWrapper.cpp) is done here with a HIP device context -- uniform blocks become one packed
// DevScene blob in HBM (rt_pack.h), textures become RGBA8 arrays, glDrawArrays becomes a kernel
// launch on the context's stream, bracketed by HIP events for the per-draw time.
// There is no CPU fallback: without a usable HIP device rtx_create fails.
#include <hip/hip_runtime.h>
// RCCL: types and prototypes only -- the library is dlopen-ed when a multi-device context asks for it, so a box without the RCCL
// development headers still builds the single-device library from the few declarations below (same ABI as rccl.h / nccl.h).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId*);
ncclResult_t ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int);
ncclResult_t ncclCommInitAll(ncclComm_t*, int, const int*);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclSend(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclRecv(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclAllGather(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
const char* ncclGetErrorString(ncclResult_t);
}
#endif

#include <dlfcn.h>

#include <chrono>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "rtx.h"
#include "rt_kernel.h"
#include "rt_pack.h"
#include "smaa_kernel.h"
#include "band_math.h"
#include "bands_kernel.h"
#include "rtx/smaa_tables.h"

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
    hipStream_t upload_stream = nullptr;
    int last_stage = -1;
    // one "last launch done" event per distinct stream the scene was ever read on: an upload on stream S waits for every other one
    std::vector<std::pair<hipStream_t, hipEvent_t>> launch_done;
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
    int opt_pencils = 1;  // RTX_OPT_RAY_PENCILS
    // ray-pencil masks of the device scene (built by upload_scene on its stream, right behind the copy)
    uint32_t* d_pencil = nullptr;
    size_t d_pencil_cap = 0;      // dwords
    uint32_t n_pencil = 0;
    hipEvent_t pencil_start = nullptr, pencil_stop = nullptr;
    bool pencil_timed = false;
    unsigned long long* d_counters = nullptr;
    // SMAA post-process (GLWrapper::enable_SMAA + the three passes of GLWrapper.cpp:173-204): smaa_preset < 0 = off
    int smaa_preset = -1;
    bool smaa_tables = false;
    unsigned smaa_frame = 0;
    uint32_t* d_screen = nullptr;    // RGBA8: what the reference shows in its window
    uint16_t* d_edges = nullptr;     // RG8, zero outside the listed pixels
    uint32_t* d_blend = nullptr;     // RGBA8, zero outside the listed pixels
    uint32_t* d_list = nullptr;      // edge pixels of the current frame
    uint32_t* d_smaa_count = nullptr;  // two alternating counter sets
    uint64_t* d_bits = nullptr;      // the edge texture as bit planes (smaa_kernel.h): rows (two planes: this resolve's and the previous one's) ...
    uint16_t* d_cbits = nullptr;     // ... and columns
    uint16_t* d_area = nullptr;
    uint8_t* d_search = nullptr;
    hipEvent_t smaa_start = nullptr, smaa_stop = nullptr;
    bool smaa_timed = false;
    bool screen_valid = false;       // d_screen holds a resolve of the current colour target's size (set by smaa_resolve)
    // multi-device (rtx_create_multi): the context the caller holds is rank 0 (the root, which owns the assembled frame); `peers` are
    // the contexts of ranks 1..N-1, ordinary single-device contexts that every scene / texture / option call is forwarded to.
    // A per-process rank (rtx_create_rank) is the same thing with the other ranks living in other processes: no peers, its own communicator.
    std::vector<rtx_context*> peers;
    rtx_context* owner = nullptr;      // set on a peer: its root
    int rank = 0;
    bool banded = false;               // the frame is split into interleaved bands over n_total ranks and assembled on rank 0
    int n_total = 1;                   // ranks the frame is split over
    bool per_process = false;          // rtx_create_rank: one rank per process
    bool loopback = false;             // RTX_GATHER_RCCL_LOOPBACK: rank 0's own bands travel through the transport too
    int gather_kind = RTX_GATHER_RCCL;
    int gather_targets = 3;            // bit 0: RGBA32F, bit 1: RGBA8 travel to the root (RTX_OPT_GATHER_TARGETS)
    int gather_rgb = 1;                // RTX_OPT_GATHER_RGB: the RGBA32F bands of the interleaved layout travel without their alpha (the constant 1.0f)
    void* d_rgb[2] = {nullptr, nullptr};   // [frame parity] this rank's float bands at 12 bytes per pixel, what is sent then (lazily allocated)
    hipEvent_t rgb_packed[2] = {nullptr, nullptr};   // recorded on the rank's transfer stream behind the pack kernel (peer-copy transport: the root waits for it)
    int band_rows = 8;                 // rows per band of the interleaved split: the kernel's tile height, the finest interleave
    // RTX_OPT_BAND_LAYOUT: 0 = interleaved bands (above); 1 = ONE contiguous range of rows per rank (split_start / split_rows, multiples of
    // 8), the root traces its range straight into the colour targets and the peers' ranges are received straight into place -- no landing
    // buffers, no placement pass; 2 = contiguous, and a single-process context re-balances the split from the ranks' own kernel times
    int band_layout = 0;
    std::vector<int> split_start, split_rows;
    size_t packed_cap_rows = 0;        // rows the packed buffers hold
    unsigned resplit_frame = 0;        // frame of the last re-balance (layout 2)
    void* d_packed[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [target][frame parity] this rank's packed bands, on its own device
    std::vector<void*> d_stage[2][2];  // root only: [target][frame parity][rank] landing buffers for the peers' bands, on the root's device
    ncclComm_t comm = nullptr;
    rtbands::ConfigDigest cfg_confirmed;   // rtx_create_rank: the frame configuration the ranks last agreed on (config_handshake)
    void* d_cfg = nullptr;             // its 16-byte messages: slot r = rank r's digest (root) / own digest + verdict (peer)
    hipStream_t xfer_stream = nullptr; // sends (peers) / receives + band placement (root) run here, beside the next frame's trace
    hipEvent_t traced[2] = {nullptr, nullptr}, moved[2] = {nullptr, nullptr};   // per frame parity: bands traced / buffers free again
    unsigned frame_no = 0;
    hipEvent_t gather_start = nullptr, gather_stop = nullptr;
    bool gather_timed = false;
    // timing
    hipEvent_t ev_start[EVENT_RING], ev_stop[EVENT_RING];
    int ev_head = 0, ev_pending = 0;
    float last_ms = 0.0f;
    uint32_t launches = 0;
    uint32_t last_variant = 0, last_tables = 0;   // rtx_stats.kernel_variant / candidate_tables of the last draw
    bool warned_tables = false;
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
    for (auto& ld : ctx->launch_done)
        if (ld.first != stream) HIP_TRY(hipStreamWaitEvent(stream, ld.second, 0));
    const int k = ctx->stage_next;
    ctx->stage_next ^= 1;
    HIP_TRY(hipEventSynchronize(ctx->stage_done[k]));  // staging buffer k is free again
    std::memcpy(ctx->h_stage[k], ctx->blob.data(), n);
    HIP_TRY(hipMemcpyAsync(ctx->d_scene, ctx->h_stage[k], n, hipMemcpyHostToDevice, stream));
    // the scene's ray pencils: masks built on the device from the blob just copied, ordered like the copy (before stage_done[k], which
    // every later launch on another stream waits for)
    const DevSceneHeader* hd = reinterpret_cast<const DevSceneHeader*>(ctx->blob.data());
    ctx->n_pencil = 0;
    ctx->pencil_timed = false;
    if (hd->n_pencil > 0 && ctx->opt_pencils) {
        if (hd->pencil_mask_words > ctx->d_pencil_cap) {
            HIP_TRY(hipDeviceSynchronize());
            if (ctx->d_pencil) HIP_TRY(hipFree(ctx->d_pencil));
            ctx->d_pencil = nullptr;
            const size_t cap = (static_cast<size_t>(hd->pencil_mask_words) + 1023) & ~static_cast<size_t>(1023);
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_pencil), cap * sizeof(uint32_t)));
            ctx->d_pencil_cap = cap;
        }
        if (!ctx->pencil_start) { HIP_TRY(hipEventCreate(&ctx->pencil_start)); HIP_TRY(hipEventCreate(&ctx->pencil_stop)); }
        HIP_TRY(hipEventRecord(ctx->pencil_start, stream));
        HIP_TRY(rt_launch_pencil_build(ctx->d_scene, *hd, reinterpret_cast<const DevPencil*>(ctx->blob.data() + hd->off_pencil), ctx->d_pencil, stream));
        HIP_TRY(hipEventRecord(ctx->pencil_stop, stream));
        ctx->n_pencil = hd->n_pencil;
        ctx->pencil_timed = true;
    }
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
            T.sky.levels = ctx->opt_lod ? it->second.levels : 1;   // RTX_OPT_TEXTURE_LOD = 0: level 0 everywhere, as for the 2-D textures
            {   // level L of the cube = six faces of max(1, size >> L)^2 dwords behind level L - 1 (rtx_cubemap_create)
                uint32_t off = 0;
                int wl = it->second.width;
                for (int l = 0; l < MAX_MIPS; l++) {
                    T.sky.level_off[l] = off;
                    off += 6u * static_cast<uint32_t>(wl) * static_cast<uint32_t>(wl);
                    wl = wl > 1 ? wl >> 1 : 1;
                }
            }
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

int draw_impl(rtx_context* ctx, int band_rows, int band_first, int band_stride, float* out_f32, uint32_t* out_u8, hipStream_t stream, int rows_limit = -1)
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
    if (rows_limit >= 0 && rows_limit < rows_local) rows_local = rows_limit;   // a contiguous range: 8-row bands from band_first on, this many rows
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
    // Rows that show a torus first: for a rank's share of a frame (band_stride >= 4) and, round 4, for any launch of at most HOT_MAX_WG
    // workgroups -- a small frame ends with its longest workgroups however few the others are (default scene, kernel us without / with:
    // 1280x720 197 / 166, 1920x1080 230 / 202, 2560x1440 318 / 287, 3200x1800 372 / 363; 3840x2160 469 / 489: there plain row order wins,
    // neighbouring rows run the same code). profiles/r04_hot_rows_small_frames.txt
    constexpr int HOT_MAX_WG = 24000;
    const int launch_wg = ((ctx->width + 31) / 32) * ((rows_local + 7) / 8);
    if (ctx->opt_hot && !ctx->opt_xcd && (band_stride >= 4 || launch_wg <= HOT_MAX_WG)) hot_rows(ctx, p);
    p.out_f32 = out_f32;
    p.out_u8 = out_u8;
    p.counters = ctx->d_counters;
    p.pencil_masks = ctx->opt_pencils && ctx->n_pencil > 0 ? ctx->d_pencil : nullptr;
    fill_tex_table(ctx, p.tex);
    if (ctx->opt_lds && p.tex.lod && p.tex.sky.levels > 1)
        return fail(RTX_ERR_INVALID, "RTX_OPT_SCENE_IN_LDS = 1 is not built for a mip-mapped sky box (rtx_cubemap_create with gen_mipmap = 1): the kernels that sample "
                                     "cube mips exist for the scalar-load scene tables only");
    if (ctx->opt_count) HIP_TRY(hipMemsetAsync(ctx->d_counters, 0, 4 * sizeof(unsigned long long), stream));
    if (ctx->ev_pending == EVENT_RING) {  // ring full: retire the oldest pair only (recorded EVENT_RING launches ago, long finished)
        const int idx = (ctx->ev_head - ctx->ev_pending + EVENT_RING * 2) % EVENT_RING;
        HIP_TRY(hipEventSynchronize(ctx->ev_stop[idx]));
        ctx->ev_pending--;
    }
    const int e = ctx->ev_head;
    static const bool marker_events = getenv("RTX_MARKER_EVENTS") != nullptr;   // A/B: hipEventRecord around the launch, as until round 4
    if (marker_events) HIP_TRY(hipEventRecord(ctx->ev_start[e], stream));
    // long primitive tables (quadric-/torus-heavy scenes): the 7-waves-per-SIMD build of the kernel hides the table walks
    const rtpack::Defines& df = ctx->defines;
    const int n_prims = df.sphere_size + df.plane_size + df.surface_size + df.box_size + df.torus_size + df.ring_size;
    // (a scene whose packer built ray pencils -- 16 .. 128 quadrics or tori -- takes the many-primitive build whatever its total: the
    // default build has no code that reads the tables)
    const bool high_occ = ctx->opt_occ < 0 ? (n_prims >= 32 || (ctx->opt_pencils && ctx->n_pencil > 0)) : ctx->opt_occ != 0;
    ctx->last_variant = high_occ ? 1u : 0u;
    ctx->last_tables = 0u;
    if (high_occ && ctx->opt_cull) {
        const DevSceneHeader* hd = reinterpret_cast<const DevSceneHeader*>(ctx->blob.data());
        if (df.surface_size >= 16 || df.torus_size >= 16) ctx->last_tables |= 1u;
        if (p.pencil_masks) ctx->last_tables |= 2u;
        if (p.pencil_masks && ctx->blob.size() >= sizeof(DevSceneHeader) && hd->off_slabs != 0u) ctx->last_tables |= 4u;
    }
    if ((df.surface_size > RT_PENCIL_MAX_PRIMS || df.torus_size > RT_PENCIL_MAX_PRIMS) && !ctx->warned_tables) {
        ctx->warned_tables = true;   // once per context: the frame is right, but this scene is outside what the candidate tables hold
        std::fprintf(stderr, "rtx: %d quadrics / %d tori: more than the %d of a kind the ray-pencil and slab tables hold -- two-level scans with group culls only "
                             "(same pixels, slower; rtx_stats.candidate_tables)\n", df.surface_size, df.torus_size, (int)RT_PENCIL_MAX_PRIMS);
    }
    if (marker_events) {
        HIP_TRY(rt_launch_trace(p, ctx->opt_cull != 0, ctx->opt_count != 0, ctx->opt_lds != 0, high_occ, stream));
        HIP_TRY(hipEventRecord(ctx->ev_stop[e], stream));
    } else {
        HIP_TRY(rt_launch_trace(p, ctx->opt_cull != 0, ctx->opt_count != 0, ctx->opt_lds != 0, high_occ, stream, ctx->ev_start[e], ctx->ev_stop[e]));
    }
    // "The last launch on this stream has finished", for whoever overwrites the scene from another stream (upload_scene): the launch's own
    // stop event, not a third event per draw (round 4: an event record costs the queue ~4 us per draw -- ms_per_step 0.4788 -> 0.4745). The
    // ring re-records an event only EVENT_RING launches later and waits for its old recording first (above), so a stream's entry never
    // names an unfinished launch other than its latest.
    {
        bool known = false;
        for (auto& ld : ctx->launch_done)
            if (ld.first == stream) { ld.second = ctx->ev_stop[e]; known = true; }
        if (!known) {
            if (ctx->launch_done.size() >= 16) {   // a caller cycling through many streams: the oldest stream's last launch is waited for here
                HIP_TRY(hipEventSynchronize(ctx->launch_done.front().second));
                ctx->launch_done.erase(ctx->launch_done.begin());
            }
            ctx->launch_done.emplace_back(stream, ctx->ev_stop[e]);
        }
    }
    ctx->ev_head = (ctx->ev_head + 1) % EVENT_RING;
    ctx->ev_pending++;
    ctx->launches++;
    return RTX_OK;
}

// The two SMAA look-up tables, generated once per process from their published construction (include/rtx/smaa_tables.h).
void default_smaa_tables(const uint8_t** area, const uint8_t** search)
{
    static std::vector<uint8_t> a, s;
    static std::once_flag once;
    std::call_once(once, [] {
        a.resize(rtx_smaa::AREA_BYTES);
        s.resize(rtx_smaa::SEARCH_BYTES);
        rtx_smaa::generate_area_table(a.data());
        rtx_smaa::generate_search_table(s.data());
    });
    *area = a.data();
    *search = s.data();
}

int upload_smaa_tables(rtx_context* ctx, const uint8_t* area_rg8, const uint8_t* search_r8)
{
    int st = use_device(ctx);
    if (st) return st;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (!ctx->d_area) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_area), rtx_smaa::AREA_BYTES));
    if (!ctx->d_search) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_search), rtx_smaa::SEARCH_BYTES));
    HIP_TRY(hipMemcpy(ctx->d_area, area_rg8, rtx_smaa::AREA_BYTES, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_search, search_r8, rtx_smaa::SEARCH_BYTES, hipMemcpyHostToDevice));
    ctx->smaa_tables = true;
    return RTX_OK;
}

int smaa_alloc(rtx_context* ctx)
{
    if (ctx->d_screen) return RTX_OK;
    const size_t px = static_cast<size_t>(ctx->width) * ctx->height;
    if (px >= (size_t(1) << 30)) return fail(RTX_ERR_INVALID, "SMAA: frames of 2^30 pixels or more are not supported (%d x %d)", ctx->width, ctx->height);   // edge-list entries: 30 bits of pixel index
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_screen), px * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_edges), px * 2));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_blend), px * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_list), smaa_segment_capacity(ctx->width, ctx->height) * SMAA_SEGMENTS * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_smaa_count), 2 * SMAA_COUNT_SET * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_bits), 2 * smaa_plane_bytes(ctx->width, ctx->height)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_cbits), smaa_col_plane_bytes(ctx->width, ctx->height)));
    HIP_TRY(hipMemsetAsync(ctx->d_bits, 0, 2 * smaa_plane_bytes(ctx->width, ctx->height), ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->d_cbits, 0, smaa_col_plane_bytes(ctx->width, ctx->height), ctx->stream));
    // Neither texture is ever cleared by the passes: the kernels read edges from the bit planes, and a weight texel only where the plane has
    // an edge pixel (smaa_kernel.hip). RTX_SMAA_POISON=1 (tests) starts the weight texture full of garbage to prove exactly that.
    const char* poison = std::getenv("RTX_SMAA_POISON");
    HIP_TRY(hipMemsetAsync(ctx->d_edges, 0, px * 2, ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->d_blend, (poison && poison[0] == '1') ? 0xa5 : 0, px * 4, ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->d_smaa_count, 0, 2 * SMAA_COUNT_SET * sizeof(uint32_t), ctx->stream));
    HIP_TRY(hipEventCreate(&ctx->smaa_start));
    HIP_TRY(hipEventCreate(&ctx->smaa_stop));
    ctx->smaa_frame = 0;
    return RTX_OK;
}

// `frame`: the resolve whose buffers are meant (the row bit plane alternates between two buffers)
SmaaBuffers smaa_buffers(rtx_context* ctx, unsigned frame)
{
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
    {   // the row plane alternates between two buffers: resolve f writes plane f & 1 and reads what resolve f - 1 wrote
        const size_t words = smaa_plane_bytes(ctx->width, ctx->height) / 8;
        b.bits = ctx->d_bits + (frame & 1u) * words;
        b.bits_prev = ctx->d_bits + ((frame & 1u) ^ 1u) * words;
    }
    b.cbits = ctx->d_cbits;
    b.area = ctx->d_area;
    b.search = ctx->d_search;
    return b;
}

// The three passes after the tracer (GLWrapper.cpp:173-204) on the context's RGBA8 colour target, into the screen buffer.
int smaa_resolve(rtx_context* ctx, hipStream_t stream)
{
    int st = smaa_alloc(ctx);
    if (st) return st;
    if (!ctx->smaa_tables) {   // the caller supplied none: the library's own (== the arrays the reference uploads, rtx/smaa_tables.h)
        const uint8_t *area = nullptr, *search = nullptr;
        default_smaa_tables(&area, &search);
        st = upload_smaa_tables(ctx, area, search);
        if (st) return st;
    }
    if (stream != ctx->stream) return fail(RTX_ERR_INVALID, "the SMAA resolve runs on the context's own stream");
    const SmaaBuffers b = smaa_buffers(ctx, ctx->smaa_frame);
    HIP_TRY(smaa_launch(b, ctx->smaa_preset, ctx->smaa_frame, stream, ctx->smaa_start, ctx->smaa_stop));   // the kernels' own timestamps
    ctx->smaa_frame++;
    ctx->smaa_timed = true;
    ctx->screen_valid = true;
    return RTX_OK;
}

// ---- multi-device contexts (rtx_create_multi) ----------------------------------------------------------------------------------
// RCCL is loaded on demand: a single-device program, and a box without the library, never touch it.
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;   // the ranks' first contact only (config_handshake); absent in a library: the check is skipped
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool load(std::string& err)
    {
        if (lib) return true;
        for (const char* name : {"librccl.so", "librccl.so.1"}) {   // a copy the process already holds (PyTorch ships one) is found first
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { const char* e = dlerror(); err = std::string("librccl.so could not be loaded: ") + (e ? e : "?"); return false; }
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        Send = reinterpret_cast<decltype(Send)>(dlsym(lib, "ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(dlsym(lib, "ncclRecv"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!CommInitAll || !CommInitRank || !GetUniqueId || !CommDestroy || !Send || !Recv || !GroupStart || !GroupEnd || !GetErrorString) {
            err = "librccl.so lacks an expected symbol";
            dlclose(lib);
            lib = nullptr;
            return false;
        }
        return true;
    }
};
Rccl g_rccl;
static_assert(sizeof(ncclUniqueId) == RTX_RCCL_ID_BYTES, "rtx.h documents the size of an RCCL unique id");
#define NCCL_TRY(expr)                                                                                          \
    do {                                                                                                        \
        ncclResult_t _r = (expr);                                                                               \
        if (_r != ncclSuccess) return fail(RTX_ERR_DEVICE, "%s failed: %s", #expr, g_rccl.GetErrorString(_r)); \
    } while (0)
// inside an open ncclGroupStart: close the group before reporting (a group left open would swallow every later RCCL call of the thread)
#define NCCL_TRY_IN_GROUP(expr)                                                                                 \
    do {                                                                                                        \
        ncclResult_t _r = (expr);                                                                               \
        if (_r != ncclSuccess) {                                                                                \
            (void)g_rccl.GroupEnd();                                                                            \
            return fail(RTX_ERR_DEVICE, "%s failed: %s", #expr, g_rccl.GetErrorString(_r));                    \
        }                                                                                                       \
    } while (0)

inline int n_ranks(const rtx_context* ctx) { return ctx->n_total; }
inline rtx_context* rank_ctx(rtx_context* ctx, int r) { return r == 0 ? ctx : ctx->peers[r - 1]; }
inline size_t target_bytes(int t) { return rtbands::target_bytes(t); }   // target 0 = RGBA32F, 1 = RGBA8 (band_math.h)
inline bool uses_rccl(const rtx_context* ctx) { return ctx->gather_kind != RTX_GATHER_PEER_COPY; }

// ---- waits that cannot hang the process, and the ranks' first contact ----------------------------------------------
// RTX_GATHER_TIMEOUT_MS (default 30 000; 0 = wait for ever, the behaviour of rounds 1-5): how long a wait on a transfer stream may take. A
// band whose peer never issued the matching ncclSend / ncclRecv -- or issued it with other byte counts -- never completes; the wait then
// ends with RTX_ERR_DEVICE and a message instead of a hung process.
double gather_timeout_ms()
{
    static const double v = [] {
        const char* e = std::getenv("RTX_GATHER_TIMEOUT_MS");
        return e && *e ? std::atof(e) : 30000.0;
    }();
    return v;
}
int wait_stream_bounded(rtx_context* c, hipStream_t s, const char* what, int peer = -1)
{
    if (!s) return RTX_OK;
    if (!c->banded || !uses_rccl(c)) { HIP_TRY(hipStreamSynchronize(s)); return RTX_OK; }   // nothing on it waits for another process
    hipError_t bad = hipSuccess;
    int polls = 0;
    const auto t0 = std::chrono::steady_clock::now();
    const bool ok = rtbands::bounded_wait(
        [&] { const hipError_t e = hipStreamQuery(s); if (e == hipSuccess) return true; if (e != hipErrorNotReady) { bad = e; return true; } return false; },
        [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); },
        [&] { if (++polls > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50)); },
        gather_timeout_ms());
    if (bad != hipSuccess) return fail(RTX_ERR_DEVICE, "%s: %s", what, hipGetErrorString(bad));
    if (ok) return RTX_OK;
    if (peer >= 0)
        return fail(RTX_ERR_DEVICE, "%s: rank %d did not answer rank %d within %.0f ms (RTX_GATHER_TIMEOUT_MS) -- is every rank making the same calls in the same order?",
                    what, peer, c->rank, gather_timeout_ms());
    return fail(RTX_ERR_DEVICE, "%s: the transfer stream of rank %d (of %d) did not finish within %.0f ms (RTX_GATHER_TIMEOUT_MS): a peer has not issued the matching "
                "ncclSend / ncclRecv, or issued it with other byte counts (band layout, split, RTX_OPT_GATHER_RGB and RTX_OPT_GATHER_TARGETS must be set on EVERY rank)",
                what, c->rank, c->n_total, gather_timeout_ms());
}
rtbands::FrameConfig frame_config(const rtx_context* me)
{
    rtbands::FrameConfig fc;
    fc.width = me->width; fc.height = me->height; fc.n_ranks = me->n_total; fc.band_rows = me->band_rows; fc.band_layout = me->band_layout;
    fc.gather_targets = me->gather_targets; fc.gather_rgb = me->gather_rgb; fc.loopback = me->loopback ? 1 : 0;
    if (me->band_layout != 0) fc.split_rows = me->split_rows;
    return fc;
}
// Ranks in separate processes: before a band travels under a configuration this rank has not had confirmed, the ranks compare 16-byte
// digests of it (band_math.h). ONE ncclAllGather of 16 bytes per rank on the transfer stream: a collective runs on the rings the communicator
// built at ncclCommInitRank, so it needs no connection of its own (a send / receive pair in each direction would: RCCL connects peers
// lazily, both directions of a pair in one blocking exchange, and a root that first receives and then answers never gets there), every rank
// sees every digest and fails by itself with the same message, and the message sizes are fixed whatever the configurations are. A rank that
// does not take part shows up as the bounded wait's timeout on the others. Costs one small collective, only when something changed.
int config_handshake(rtx_context* me)
{
    if (!me->per_process || !me->banded || !uses_rccl(me) || !g_rccl.AllGather) return RTX_OK;
    const rtbands::ConfigDigest mine = rtbands::config_digest(frame_config(me));
    if (mine == me->cfg_confirmed) return RTX_OK;
    int st = use_device(me);
    if (st) return st;
    const int N = me->n_total;
    const size_t slot = 16;
    if (!me->d_cfg) HIP_TRY(hipMalloc(&me->d_cfg, slot * static_cast<size_t>(N + 1)));
    auto at = [&](int k) { return static_cast<void*>(static_cast<char*>(me->d_cfg) + slot * static_cast<size_t>(k)); };
    unsigned long long msg[2] = {mine.a, mine.b};
    HIP_TRY(hipMemcpyAsync(at(N), msg, slot, hipMemcpyHostToDevice, me->xfer_stream));            // own digest behind the N gathered slots
    NCCL_TRY(g_rccl.AllGather(at(N), at(0), slot, ncclUint8, me->comm, me->xfer_stream));
    if ((st = wait_stream_bounded(me, me->xfer_stream, "frame-configuration check (is every rank drawing, with the same calls in the same order?)")) != RTX_OK) return st;
    std::vector<unsigned long long> host(static_cast<size_t>(2 * N));
    HIP_TRY(hipMemcpy(host.data(), me->d_cfg, slot * static_cast<size_t>(N), hipMemcpyDeviceToHost));
    std::vector<rtbands::ConfigDigest> all(static_cast<size_t>(N));
    for (int r = 0; r < N; r++) { all[r].a = host[2 * r]; all[r].b = host[2 * r + 1]; }
    if (all[me->rank] != mine) return fail(RTX_ERR_DEVICE, "frame-configuration check: rank %d did not get its own digest back from the all-gather", me->rank);
    const int wrong = rtbands::config_first_mismatch(all);
    if (wrong >= 0)
        return fail(RTX_ERR_INVALID, "frame configuration of rank %d differs from rank 0's (this is rank %d): frame size, rank count, RTX_OPT_BAND_LAYOUT, the band split, "
                    "RTX_OPT_GATHER_TARGETS and RTX_OPT_GATHER_RGB must be the same on every rank", wrong, me->rank);
    me->cfg_confirmed = mine;
    return RTX_OK;
}

int rows_of_rank(const rtx_context* root, int rank)
{
    if (root->band_layout != 0 && rank < static_cast<int>(root->split_rows.size())) return root->split_rows[rank];
    return rtbands::rows_interleaved(root->height, root->band_rows, n_ranks(root), rank);
}

// ---- contiguous bands (RTX_OPT_BAND_LAYOUT 1 / 2) ------------------------------------------------------------------------------
// One range of rows per rank instead of interleaved bands: what a rank traces is a sub-frame, so the root traces its range straight into
// the colour targets and receives every peer's range straight into its place -- no landing buffers and no placement kernel (at 8K RGBA32F
// that pass re-copied 464 MB per frame on the root, VERDICT r3 weak #6d). The price is balance: sky rows cost a tenth of object rows, so
// the split is weighted -- by the caller (rtx_set_band_split: every rank of a per-process group names the same split) or, in a
// single-process context with layout 2, by the library from the ranks' own kernel times two frames back.
void split_equal(rtx_context* c) { rtbands::split_equal(c->height, n_ranks(c), c->split_rows, c->split_start); }
// rows[r] rows for rank r, in rank order from row 0: every count a multiple of 8 except the last non-empty one, together the frame
int split_set(rtx_context* c, const int* rows, int n)
{
    const int N = n_ranks(c);
    int bad = 0;
    long long total = 0;
    switch (rtbands::split_check(c->height, rows, n, N, &bad, &total)) {
        case 0: break;
        case 1: return fail(RTX_ERR_INVALID, "rtx_set_band_split: %d counts for %d ranks", n, N);
        case 2: return fail(RTX_ERR_INVALID, "rtx_set_band_split: negative row count");
        case 3: return fail(RTX_ERR_INVALID, "rtx_set_band_split: rank %d gets %d rows -- ranges must start on a multiple of 8 (the kernel's tile height)", bad, rows[bad]);
        default: return fail(RTX_ERR_INVALID, "rtx_set_band_split: the ranges cover %lld rows, the frame has %d", total, c->height);
    }
    c->split_rows.assign(rows, rows + N);
    rtbands::starts_of(c->split_rows, c->split_start);
    return RTX_OK;
}
// a rank's packed buffers must hold whatever range a re-split may hand it: the whole frame
int ensure_packed(rtx_context* c, const rtx_context* frame)
{
    const size_t need = static_cast<size_t>(frame->height) + 8;
    if (c->packed_cap_rows >= need) return RTX_OK;
    int st = use_device(c);
    if (st) return st;
    HIP_TRY(hipDeviceSynchronize());
    for (int t = 0; t < 2; t++)
        for (int q = 0; q < 2; q++) {
            if (c->d_packed[t][q]) HIP_TRY(hipFree(c->d_packed[t][q]));
            c->d_packed[t][q] = nullptr;
            HIP_TRY(hipMalloc(&c->d_packed[t][q], need * frame->width * target_bytes(t)));
        }
    for (int q = 0; q < 2; q++)
        if (c->d_rgb[q]) { HIP_TRY(hipFree(c->d_rgb[q])); c->d_rgb[q] = nullptr; HIP_TRY(hipEventDestroy(c->rgb_packed[q])); c->rgb_packed[q] = nullptr; }   // re-made at the new size
    c->packed_cap_rows = need;
    return RTX_OK;
}
// kernel time of the launch `ago` launches back on this rank, if it has finished (never blocks)
bool launch_ms_ago(rtx_context* c, int ago, float* ms)
{
    if (ago < 1 || ago > EVENT_RING - 1 || static_cast<int>(c->launches) < ago) return false;
    const int idx = (c->ev_head - ago + EVENT_RING * 2) % EVENT_RING;
    if (hipSetDevice(c->device) != hipSuccess || hipEventQuery(c->ev_stop[idx]) != hipSuccess) return false;
    return hipEventElapsedTime(ms, c->ev_start[idx], c->ev_stop[idx]) == hipSuccess;
}
// layout 2, single process: move the boundaries towards equal kernel times, from the ranks' own kernel times two frames back (those launches
// have finished: nothing waits). The arithmetic -- and its guard against frames with fewer 8-row units than ranks -- is band_math.h rebalance.
void rebalance(rtx_context* root)
{
    const int N = n_ranks(root);
    if (root->per_process || N < 2 || root->frame_no < 4 || root->frame_no - root->resplit_frame < 3) return;
    std::vector<double> ms(N, 0.0);
    for (int r = 0; r < N; r++) {
        if (root->split_rows[r] <= 0) continue;             // a rank without rows has no time of its own
        float t = 0.0f;
        if (!launch_ms_ago(rank_ctx(root, r), 2, &t) || !(t > 0.0f)) return;
        ms[r] = t;
    }
    std::vector<int> rows, start;
    if (!rtbands::rebalance(root->height, root->split_rows, ms, rows, start)) return;
    root->split_rows.swap(rows);
    root->split_start.swap(start);
    root->resplit_frame = root->frame_no;
}
int multi_draw_contiguous(rtx_context* me)
{
    const int N = n_ranks(me), par = rtbands::buffer_set(me->frame_no);
    const bool root_here = me->rank == 0;
    const int first_moved = me->loopback ? 0 : 1;
    std::vector<rtx_context*> local;
    if (me->per_process) local.push_back(me);
    else for (int r = 0; r < N; r++) local.push_back(rank_ctx(me, r));
    int st;
    if (static_cast<int>(me->split_rows.size()) != N) split_equal(me);
    if (me->band_layout == 2) rebalance(me);
    if ((st = config_handshake(me)) != RTX_OK) return st;
    for (rtx_context* c : local)
        if (c->rank >= first_moved && (st = ensure_packed(c, me)) != RTX_OK) return st;
    const size_t W = static_cast<size_t>(me->width);
    auto in_place = [&](int t, int r) -> void* {            // where rank r's range lies in the root's colour target t
        const size_t off = static_cast<size_t>(me->split_start[r]) * W;
        return t == 0 ? static_cast<void*>(me->d_fb_f32 + off * 4) : static_cast<void*>(me->d_fb_u8 + off);
    };
    for (rtx_context* c : local) {
        if ((st = use_device(c)) != RTX_OK) return st;
        const bool direct = c->rank < first_moved;           // the root's own range (unless it travels too: loopback)
        if (me->frame_no >= 2 && !direct) HIP_TRY(hipStreamWaitEvent(c->stream, c->moved[par], 0));
        float* of = nullptr;
        uint32_t* ou = nullptr;
        if (me->gather_targets & 1) of = direct ? static_cast<float*>(in_place(0, c->rank)) : static_cast<float*>(c->d_packed[0][par]);
        if (me->gather_targets & 2) ou = direct ? static_cast<uint32_t*>(in_place(1, c->rank)) : static_cast<uint32_t*>(c->d_packed[1][par]);
        // after a re-split the root's range may reach into rows the previous frame's receives are still filling: that gather must be complete
        // first (only then: ordinarily the root's trace of frame k overlaps the gather of frame k-1)
        if (direct && me->frame_no >= 1 && me->resplit_frame == me->frame_no) HIP_TRY(hipStreamWaitEvent(c->stream, me->moved[(me->frame_no - 1u) & 1u], 0));
        st = draw_impl(c, 8, me->split_start[c->rank] / 8, 1, of, ou, c->stream, me->split_rows[c->rank]);
        if (st) return st;
        HIP_TRY(hipEventRecord(c->traced[par], c->stream));
        HIP_TRY(hipStreamWaitEvent(c->xfer_stream, c->traced[par], 0));
    }
    if (root_here) {
        if ((st = use_device(me)) != RTX_OK) return st;
        HIP_TRY(hipEventRecord(me->gather_start, me->xfer_stream));
    }
    if (uses_rccl(me)) {
        if (N > first_moved) {
            NCCL_TRY(g_rccl.GroupStart());
            for (rtx_context* c : local) {
                if (c->rank < first_moved) continue;
                const size_t rows = static_cast<size_t>(me->split_rows[c->rank]);
                for (int t = 0; t < 2; t++)
                    if (((me->gather_targets >> t) & 1) && rows)
                        NCCL_TRY_IN_GROUP(g_rccl.Send(c->d_packed[t][par], rows * W * target_bytes(t), ncclUint8, 0, c->comm, c->xfer_stream));
            }
            if (root_here)
                for (int r = first_moved; r < N; r++) {
                    const size_t rows = static_cast<size_t>(me->split_rows[r]);
                    for (int t = 0; t < 2; t++)
                        if (((me->gather_targets >> t) & 1) && rows)
                            NCCL_TRY_IN_GROUP(g_rccl.Recv(in_place(t, r), rows * W * target_bytes(t), ncclUint8, r, me->comm, me->xfer_stream));
                }
            NCCL_TRY(g_rccl.GroupEnd());
        }
    } else {
        for (int r = 1; r < N; r++) {
            rtx_context* c = rank_ctx(me, r);
            HIP_TRY(hipStreamWaitEvent(me->xfer_stream, c->traced[par], 0));
            const size_t rows = static_cast<size_t>(me->split_rows[r]);
            for (int t = 0; t < 2; t++)
                if (((me->gather_targets >> t) & 1) && rows)
                    HIP_TRY(hipMemcpyPeerAsync(in_place(t, r), me->device, c->d_packed[t][par], c->device, rows * W * target_bytes(t), me->xfer_stream));
        }
    }
    if (root_here) {
        HIP_TRY(hipEventRecord(me->gather_stop, me->xfer_stream));
        me->gather_timed = true;
    }
    for (rtx_context* c : local) {
        if (c->rank == 0 || uses_rccl(me)) {
            if ((st = use_device(c)) != RTX_OK) return st;
            HIP_TRY(hipEventRecord(c->moved[par], c->xfer_stream));
        } else {
            c->moved[par] = me->moved[par];
        }
    }
    me->frame_no++;
    return RTX_OK;
}

// GLWrapper::draw on N devices (BASELINE north_star: "GLWrapper dispatch -> HIP launch + RCCL tile gather"). Every rank traces its
// interleaved row bands into packed buffers on its own device and stream (one launch writes the colour targets that travel); the peers'
// packed bands travel to the root -- one ncclSend / ncclRecv pair per peer and target inside ONE group, i.e. every peer straight over its
// own xGMI link, no ring -- and a copy kernel puts every rank's rows in their place in the root's colour targets. Sends, receives and
// placement run on each device's transfer stream and the packed / landing buffers alternate between two sets, so the gather of frame k
// overlaps the trace of frame k+1 on all devices. RTX_GATHER_PEER_COPY replaces the RCCL pair by hipMemcpyPeerAsync on the root's transfer
// stream (same data path over xGMI, no library; also the only mode in which two ranks may share a device, which the tests use on
// single-GPU boxes). `me` is the root of a single-process group (the loops then run over all ranks) or the one rank this process owns
// (rtx_create_rank: the same calls, every process issuing its own share -- sends on a peer, the receives and the placement on rank 0).
int multi_draw(rtx_context* me)
{
    if (me->band_layout != 0) return multi_draw_contiguous(me);
    { const int hs = config_handshake(me); if (hs) return hs; }
    const int N = n_ranks(me), par = rtbands::buffer_set(me->frame_no);
    const bool root_here = me->rank == 0;
    const int first_moved = me->loopback ? 0 : 1;   // first rank whose bands go through the transport
    std::vector<rtx_context*> local;                 // the ranks this process drives
    if (me->per_process) local.push_back(me);
    else for (int r = 0; r < N; r++) local.push_back(rank_ctx(me, r));
    int st;
    for (rtx_context* c : local) {
        if ((st = use_device(c)) != RTX_OK) return st;
        if (me->frame_no >= 2) HIP_TRY(hipStreamWaitEvent(c->stream, c->moved[par], 0));   // this buffer set: its previous transfer is done
        st = draw_impl(c, me->band_rows, c->rank, N, (me->gather_targets & 1) ? static_cast<float*>(c->d_packed[0][par]) : nullptr,
                       (me->gather_targets & 2) ? static_cast<uint32_t*>(c->d_packed[1][par]) : nullptr, c->stream);
        if (st) return st;
        HIP_TRY(hipEventRecord(c->traced[par], c->stream));
        HIP_TRY(hipStreamWaitEvent(c->xfer_stream, c->traced[par], 0));
    }
    if (root_here) {
        if ((st = use_device(me)) != RTX_OK) return st;
        HIP_TRY(hipEventRecord(me->gather_start, me->xfer_stream));
    }
    // The float target without its alpha: every rank whose bands travel packs them to 12 bytes per pixel on its transfer stream (behind its
    // trace, beside the next one), and that is what is sent; the root writes the 1.0f back while it places the bands.
    const bool rgb = me->gather_rgb != 0 && (me->gather_targets & 1) != 0;
    auto bytes_moved = [&](int t, size_t rows) { return rtbands::bytes_moved(me->width, rows, t, rgb); };
    if (rgb)
        for (rtx_context* c : local) {
            if (c->rank < first_moved) continue;
            if ((st = use_device(c)) != RTX_OK) return st;
            const size_t rows = static_cast<size_t>(rows_of_rank(me, c->rank));
            for (int q = 0; q < 2; q++)
                if (!c->d_rgb[q]) {
                    HIP_TRY(hipMalloc(&c->d_rgb[q], (c->packed_cap_rows ? c->packed_cap_rows : rows + 8) * me->width * 12));
                    HIP_TRY(hipEventCreateWithFlags(&c->rgb_packed[q], hipEventDisableTiming));
                }
            HIP_TRY(bands_pack_rgb(c->d_packed[0][par], c->d_rgb[par], rows * me->width, c->xfer_stream));
            HIP_TRY(hipEventRecord(c->rgb_packed[par], c->xfer_stream));
        }
    if (rgb && root_here && (st = use_device(me)) != RTX_OK) return st;   // what follows on the root is issued with the root's device current, as before
    auto send_buffer = [&](rtx_context* c, int t) { return t == 0 && rgb ? c->d_rgb[par] : c->d_packed[t][par]; };
    if (uses_rccl(me)) {
        if (N > first_moved) {
            NCCL_TRY(g_rccl.GroupStart());
            for (rtx_context* c : local) {
                if (c->rank < first_moved) continue;
                const size_t rows = static_cast<size_t>(rows_of_rank(me, c->rank));
                for (int t = 0; t < 2; t++)
                    if ((me->gather_targets >> t) & 1)
                        NCCL_TRY_IN_GROUP(g_rccl.Send(send_buffer(c, t), bytes_moved(t, rows), ncclUint8, 0, c->comm, c->xfer_stream));
            }
            if (root_here)
                for (int r = first_moved; r < N; r++) {
                    const size_t rows = static_cast<size_t>(rows_of_rank(me, r));
                    for (int t = 0; t < 2; t++)
                        if ((me->gather_targets >> t) & 1)
                            NCCL_TRY_IN_GROUP(g_rccl.Recv(me->d_stage[t][par][r], bytes_moved(t, rows), ncclUint8, r, me->comm, me->xfer_stream));
                }
            NCCL_TRY(g_rccl.GroupEnd());
        }
    } else {
        for (int r = 1; r < N; r++) {
            rtx_context* c = rank_ctx(me, r);
            HIP_TRY(hipStreamWaitEvent(me->xfer_stream, rgb ? c->rgb_packed[par] : c->traced[par], 0));
            const size_t rows = static_cast<size_t>(rows_of_rank(me, r));
            for (int t = 0; t < 2; t++) {
                if (!((me->gather_targets >> t) & 1)) continue;
                HIP_TRY(hipMemcpyPeerAsync(me->d_stage[t][par][r], me->device, send_buffer(c, t), c->device, bytes_moved(t, rows), me->xfer_stream));
            }
        }
    }
    if (root_here) {
        for (int r = 0; r < N; r++)
            for (int t = 0; t < 2; t++) {
                if (!((me->gather_targets >> t) & 1)) continue;
                const void* src = r < first_moved ? me->d_packed[t][par] : me->d_stage[t][par][r];
                void* dst = t == 0 ? static_cast<void*>(me->d_fb_f32) : static_cast<void*>(me->d_fb_u8);
                if (t == 0 && rgb && r >= first_moved) HIP_TRY(bands_unpack_rgb(src, dst, me->width, me->height, me->band_rows, r, N, rows_of_rank(me, r), me->xfer_stream));
                else HIP_TRY(bands_unpack(src, dst, me->width, me->height, static_cast<int>(target_bytes(t)), me->band_rows, r, N, rows_of_rank(me, r), me->xfer_stream));
            }
        HIP_TRY(hipEventRecord(me->gather_stop, me->xfer_stream));
        me->gather_timed = true;
    }
    for (rtx_context* c : local) {   // a rank's packed buffers are free once its send (RCCL) / the root's copy and placement have completed
        if (c->rank == 0 || uses_rccl(me)) {
            if ((st = use_device(c)) != RTX_OK) return st;
            HIP_TRY(hipEventRecord(c->moved[par], c->xfer_stream));
        } else {
            c->moved[par] = me->moved[par];   // (shared handle: recorded on the root's transfer stream above; owned by the root)
        }
    }
    me->frame_no++;
    return RTX_OK;
}

// the assembled frame is written on the root's transfer stream: whoever reads the colour targets waits for it
int multi_sync(rtx_context* root)
{
    if (!root->banded) return RTX_OK;
    int st = use_device(root);
    if (st) return st;
    return wait_stream_bounded(root, root->xfer_stream, "gather");
}

// the packed band buffers, transfer stream and events of one rank (on its own device); `frame` supplies size, rank count and transport
int rank_alloc(rtx_context* c, const rtx_context* frame, int rank)
{
    int st = use_device(c);
    if (st) return st;
    c->rank = rank;
    HIP_TRY(hipStreamCreateWithFlags(&c->xfer_stream, hipStreamNonBlocking));
    const size_t rows = static_cast<size_t>(rows_of_rank(frame, rank)) + 8;
    for (int t = 0; t < 2; t++)
        for (int p = 0; p < 2; p++) HIP_TRY(hipMalloc(&c->d_packed[t][p], rows * frame->width * target_bytes(t)));
    c->packed_cap_rows = rows;
    for (int p = 0; p < 2; p++) {
        HIP_TRY(hipEventCreateWithFlags(&c->traced[p], hipEventDisableTiming));
        if (rank == 0 || uses_rccl(frame)) HIP_TRY(hipEventCreateWithFlags(&c->moved[p], hipEventDisableTiming));
    }
    return RTX_OK;
}

// rank 0 only: landing buffers for every rank whose bands travel, and the gather's timing events
int root_alloc(rtx_context* root)
{
    const int N = n_ranks(root);
    int st = use_device(root);
    if (st) return st;
    HIP_TRY(hipEventCreate(&root->gather_start));
    HIP_TRY(hipEventCreate(&root->gather_stop));
    for (int t = 0; t < 2; t++)
        for (int p = 0; p < 2; p++) {
            root->d_stage[t][p].assign(N, nullptr);
            for (int r = root->loopback ? 0 : 1; r < N; r++)
                HIP_TRY(hipMalloc(&root->d_stage[t][p][r], (static_cast<size_t>(rows_of_rank(root, r)) + 8) * root->width * target_bytes(t)));
        }
    return RTX_OK;
}

int multi_alloc(rtx_context* root)
{
    const int N = n_ranks(root);
    for (int r = 0; r < N; r++) {
        int st = rank_alloc(rank_ctx(root, r), root, r);
        if (st) return st;
    }
    // Peer access between the root and every other device of the group, both ways: hipMemcpyPeerAsync (RTX_GATHER_PEER_COPY) then moves the
    // bands by direct DMA over the xGMI link instead of staging them through host memory (VERDICT r3 weak #6a), and RCCL's own P2P transport
    // finds the mapping in place. Devices that cannot reach each other keep the staged path -- said once, not an error.
    for (int r = 1; r < N; r++) {
        const int a = root->device, b = rank_ctx(root, r)->device;
        if (a == b) continue;
        for (int dir = 0; dir < 2; dir++) {
            const int from = dir ? b : a, to = dir ? a : b;
            int can = 0;
            if (hipSetDevice(from) != hipSuccess || hipDeviceCanAccessPeer(&can, from, to) != hipSuccess) { (void)hipGetLastError(); continue; }
            if (!can) { std::fprintf(stderr, "rtx: device %d cannot access device %d directly: band transfers between them are staged\n", from, to); continue; }
            const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) std::fprintf(stderr, "rtx: hipDeviceEnablePeerAccess(%d -> %d): %s\n", from, to, hipGetErrorString(e));
            (void)hipGetLastError();
        }
    }
    (void)hipSetDevice(root->device);
    return root_alloc(root);
}

}  // namespace

extern "C" {

// a multi-device context forwards every call that changes scene, texture or option state to its peers (replicated inputs)
#define RTX_FORWARD(call)                                             \
    do {                                                              \
        for (rtx_context * _p : ctx->peers) {                         \
            rtx_context* ctx = _p;                                    \
            const int _st = (call);                                   \
            if (_st != RTX_OK) return _st;                            \
        }                                                             \
    } while (0)

// Texture creation on a multi-device context must leave every rank with the SAME handle for the same texture, also when one rank fails
// (out of memory on one device): the texture is then removed from every rank that created it and the handle counters are brought back in
// step, so that the next creation numbers alike everywhere. (Blocks need nothing of the kind: their handles are fixed, binding slot + 1.)
static void forget_texture(rtx_context* c, uint32_t h)
{
    auto it = c->textures.find(h);
    if (it == c->textures.end()) return;
    (void)hipSetDevice(c->device);
    if (it->second.d_texels) (void)hipFree(it->second.d_texels);
    c->textures.erase(it);
}
#define RTX_FORWARD_TEXTURE(call)                                                                       \
    do {                                                                                                \
        const uint32_t _h = *handle;                                                                    \
        uint32_t* const _root_handle = handle;   /* (the loop body shadows `handle`) */                  \
        for (rtx_context * _p : ctx->peers) {                                                           \
            rtx_context* ctx = _p;                                                                      \
            uint32_t _hp = 0;                                                                           \
            uint32_t* handle = &_hp;                                                                    \
            int _st = (call);                                                                           \
            if (_st == RTX_OK && _hp != _h) _st = fail(RTX_ERR_DEVICE, "texture handles out of step across the devices (%u on the root, %u on device %d)", _h, _hp, _p->device); \
            if (_st != RTX_OK) {                                                                        \
                const std::string _msg = g_error;                                                       \
                rtx_context* _root = _p->owner;                                                         \
                forget_texture(_root, _h);                                                              \
                for (rtx_context * _q : _root->peers) { forget_texture(_q, _h); forget_texture(_q, _hp); _q->next_handle = _root->next_handle; } \
                (void)hipSetDevice(_root->device);                                                      \
                g_error = _msg;                                                                         \
                *_root_handle = 0;   /* the texture is gone on every device: the caller must not keep its number */ \
                return _st;                                                                             \
            }                                                                                           \
        }                                                                                               \
    } while (0)

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

int rtx_create_multi(int width, int height, int n_devices, const int* device_ids, int gather, rtx_context** out)
{
    if (!out || !device_ids || n_devices < 1 || n_devices > 64) return fail(RTX_ERR_INVALID, "rtx_create_multi: bad arguments");
    if (gather != RTX_GATHER_RCCL && gather != RTX_GATHER_PEER_COPY && gather != RTX_GATHER_RCCL_LOOPBACK) return fail(RTX_ERR_INVALID, "unknown gather kind %d", gather);
    *out = nullptr;
    if (gather != RTX_GATHER_PEER_COPY)
        for (int a = 0; a < n_devices; a++)
            for (int b = a + 1; b < n_devices; b++)
                if (device_ids[a] == device_ids[b]) return fail(RTX_ERR_INVALID, "device %d listed twice: RCCL needs one device per rank (RTX_GATHER_PEER_COPY allows it)", device_ids[a]);
    int have = 0;
    if (hipGetDeviceCount(&have) == hipSuccess)
        for (int a = 0; a < n_devices; a++)
            if (device_ids[a] < 0 || device_ids[a] >= have)
                return fail(RTX_ERR_INVALID, "device %d out of range: %d device%s requested, this node has %d", device_ids[a], n_devices, n_devices == 1 ? "" : "s", have);
    rtx_context* root = nullptr;
    int st = rtx_create(width, height, device_ids[0], &root);
    if (st) return st;
    root->gather_kind = gather;
    root->loopback = gather == RTX_GATHER_RCCL_LOOPBACK;
    root->n_total = n_devices;
    root->banded = n_devices > 1 || root->loopback;
    for (int r = 1; r < n_devices; r++) {
        rtx_context* peer = nullptr;
        st = rtx_create(width, height, device_ids[r], &peer);
        if (st) { rtx_destroy(root); return st; }
        peer->owner = root;
        peer->gather_kind = gather;
        root->peers.push_back(peer);
    }
    {
        std::lock_guard<std::mutex> lk(g_mutex);
        g_current = root;   // rtx_create made the last peer current
    }
    if (root->banded) {
        st = multi_alloc(root);
        if (st == RTX_OK && uses_rccl(root)) {
            std::string err;
            if (!g_rccl.load(err)) st = fail(RTX_ERR_DEVICE, "%s", err.c_str());
            if (st == RTX_OK) {
                std::vector<ncclComm_t> comms(n_devices);
                ncclResult_t r = g_rccl.CommInitAll(comms.data(), n_devices, device_ids);
                if (r != ncclSuccess) st = fail(RTX_ERR_DEVICE, "ncclCommInitAll failed: %s", g_rccl.GetErrorString(r));
                else for (int k = 0; k < n_devices; k++) rank_ctx(root, k)->comm = comms[k];
            }
        }
        if (st) { rtx_destroy(root); return st; }
    }
    *out = root;
    return RTX_OK;
}

int rtx_rccl_unique_id(uint8_t id[RTX_RCCL_ID_BYTES])
{
    if (!id) return fail(RTX_ERR_INVALID, "rtx_rccl_unique_id: null argument");
    std::string err;
    if (!g_rccl.load(err)) return fail(RTX_ERR_DEVICE, "%s", err.c_str());
    ncclUniqueId u;
    NCCL_TRY(g_rccl.GetUniqueId(&u));
    std::memcpy(id, &u, RTX_RCCL_ID_BYTES);
    return RTX_OK;
}

int rtx_create_rank(int width, int height, int device, int rank, int n_ranks_, const uint8_t id[RTX_RCCL_ID_BYTES], int gather, rtx_context** out)
{
    if (!out || !id || n_ranks_ < 1 || n_ranks_ > 4096 || rank < 0 || rank >= n_ranks_) return fail(RTX_ERR_INVALID, "rtx_create_rank: bad arguments (rank %d of %d)", rank, n_ranks_);
    if (gather != RTX_GATHER_RCCL && gather != RTX_GATHER_RCCL_LOOPBACK) return fail(RTX_ERR_INVALID, "rtx_create_rank: ranks in separate processes exchange their bands over RCCL (gather %d)", gather);
    *out = nullptr;
    rtx_context* ctx = nullptr;
    int st = rtx_create(width, height, device, &ctx);
    if (st) return st;
    ctx->gather_kind = gather;
    ctx->loopback = gather == RTX_GATHER_RCCL_LOOPBACK;
    ctx->n_total = n_ranks_;
    ctx->rank = rank;
    ctx->per_process = true;
    ctx->banded = n_ranks_ > 1 || ctx->loopback;
    if (ctx->banded) {
        st = rank_alloc(ctx, ctx, rank);
        if (st == RTX_OK && rank == 0) st = root_alloc(ctx);
        if (st == RTX_OK) {
            std::string err;
            if (!g_rccl.load(err)) st = fail(RTX_ERR_DEVICE, "%s", err.c_str());
        }
        if (st == RTX_OK) {
            ncclUniqueId u;
            std::memcpy(&u, id, RTX_RCCL_ID_BYTES);
            st = use_device(ctx);
            if (st == RTX_OK) {
                ncclResult_t r = g_rccl.CommInitRank(&ctx->comm, n_ranks_, u, rank);   // collective: returns once every rank has joined
                if (r != ncclSuccess) st = fail(RTX_ERR_DEVICE, "ncclCommInitRank(rank %d of %d) failed: %s", rank, n_ranks_, g_rccl.GetErrorString(r));
            }
        }
        if (st) { rtx_destroy(ctx); return st; }
    }
    *out = ctx;
    return RTX_OK;
}

int rtx_device_count(rtx_context* ctx, int* n)
{
    if (!ctx || !n) return fail(RTX_ERR_INVALID, "null argument");
    *n = n_ranks(ctx);
    return RTX_OK;
}

int rtx_set_band_split(rtx_context* ctx, const int* rows_per_rank, int n_ranks_)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "null context");
    if (ctx->owner) return fail(RTX_ERR_INVALID, "rtx_set_band_split on a peer of a multi-device context: call it on the root");
    if (!ctx->banded) return fail(RTX_ERR_INVALID, "rtx_set_band_split: not a multi-device context");
    int st = multi_sync(ctx);
    if (st) return st;
    st = split_set(ctx, rows_per_rank, n_ranks_);
    if (st) return st;
    ctx->resplit_frame = ctx->frame_no;
    if (ctx->band_layout == 0) ctx->band_layout = 1;
    return RTX_OK;
}
int rtx_get_band_split(rtx_context* ctx, int* rows_per_rank, int n_ranks_)
{
    if (!ctx || !rows_per_rank) return fail(RTX_ERR_INVALID, "null argument");
    if (n_ranks_ != n_ranks(ctx)) return fail(RTX_ERR_INVALID, "rtx_get_band_split: %d entries for %d ranks", n_ranks_, n_ranks(ctx));
    for (int r = 0; r < n_ranks_; r++) rows_per_rank[r] = ctx->banded ? rows_of_rank(ctx, r) : ctx->height;
    return RTX_OK;
}
int rtx_get_rank_draw_ms(rtx_context* ctx, float* ms_per_rank, int n_ranks_)
{
    if (!ctx || !ms_per_rank) return fail(RTX_ERR_INVALID, "null argument");
    if (n_ranks_ != n_ranks(ctx)) return fail(RTX_ERR_INVALID, "rtx_get_rank_draw_ms: %d entries for %d ranks", n_ranks_, n_ranks(ctx));
    for (int r = 0; r < n_ranks_; r++) ms_per_rank[r] = -1.0f;
    std::vector<rtx_context*> local;
    if (ctx->per_process || !ctx->banded) local.push_back(ctx);
    else for (int r = 0; r < n_ranks_; r++) local.push_back(rank_ctx(ctx, r));
    for (rtx_context* c : local) {
        int st = use_device(c);
        if (st == RTX_OK) st = drain_events(c);
        if (st) return st;
        ms_per_rank[c->rank] = c->last_ms;
    }
    return use_device(ctx);
}
int rtx_rank(rtx_context* ctx, int* rank)
{
    if (!ctx || !rank) return fail(RTX_ERR_INVALID, "null argument");
    *rank = ctx->rank;
    return RTX_OK;
}

void rtx_destroy(rtx_context* ctx)
{
    if (!ctx) return;
    for (rtx_context* p : ctx->peers) {
        (void)hipSetDevice(p->device);
        if (p->xfer_stream) (void)wait_stream_bounded(p, p->xfer_stream, "rtx_destroy");
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->xfer_stream) (void)wait_stream_bounded(ctx, ctx->xfer_stream, "rtx_destroy");   // (a transfer that will never complete must not hang the exit)
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (rtx_context* p : ctx->peers) {
        if (!uses_rccl(ctx)) p->moved[0] = p->moved[1] = nullptr;   // shared with the root's events in peer-copy mode
        rtx_destroy(p);
    }
    ctx->peers.clear();
    (void)hipSetDevice(ctx->device);
    if (ctx->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->comm);
    for (int t = 0; t < 2; t++)
        for (int p = 0; p < 2; p++) {
            if (ctx->d_packed[t][p]) (void)hipFree(ctx->d_packed[t][p]);
            for (void* q : ctx->d_stage[t][p])
                if (q) (void)hipFree(q);
        }
    for (int p = 0; p < 2; p++) {
        if (ctx->d_rgb[p]) (void)hipFree(ctx->d_rgb[p]);
        if (ctx->rgb_packed[p]) (void)hipEventDestroy(ctx->rgb_packed[p]);
        if (ctx->traced[p]) (void)hipEventDestroy(ctx->traced[p]);
        if (ctx->moved[p]) (void)hipEventDestroy(ctx->moved[p]);
    }
    if (ctx->d_cfg) (void)hipFree(ctx->d_cfg);
    if (ctx->gather_start) (void)hipEventDestroy(ctx->gather_start);
    if (ctx->gather_stop) (void)hipEventDestroy(ctx->gather_stop);
    if (ctx->xfer_stream) (void)hipStreamDestroy(ctx->xfer_stream);
    for (auto& kv : ctx->textures)
        if (kv.second.d_texels) (void)hipFree(kv.second.d_texels);
    if (ctx->d_scene) (void)hipFree(ctx->d_scene);
    if (ctx->d_pencil) (void)hipFree(ctx->d_pencil);
    if (ctx->pencil_start) (void)hipEventDestroy(ctx->pencil_start);
    if (ctx->pencil_stop) (void)hipEventDestroy(ctx->pencil_stop);
    for (int k = 0; k < 2; k++) {
        if (ctx->h_stage[k]) (void)hipHostFree(ctx->h_stage[k]);
        if (ctx->stage_done[k]) (void)hipEventDestroy(ctx->stage_done[k]);
    }
    ctx->launch_done.clear();   // (its events are the ring's stop events, destroyed with the ring)
    if (ctx->d_fb_f32) (void)hipFree(ctx->d_fb_f32);
    if (ctx->d_fb_u8) (void)hipFree(ctx->d_fb_u8);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    for (void* p : {static_cast<void*>(ctx->d_screen), static_cast<void*>(ctx->d_edges), static_cast<void*>(ctx->d_blend), static_cast<void*>(ctx->d_list),
                    static_cast<void*>(ctx->d_smaa_count), static_cast<void*>(ctx->d_area), static_cast<void*>(ctx->d_search), static_cast<void*>(ctx->d_bits),
                    static_cast<void*>(ctx->d_cbits)})
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
    RTX_FORWARD(rtx_specialize(ctx, d));
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
    RTX_FORWARD(rtx_block_create(ctx, name, 0, size, data, handle));
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
    RTX_FORWARD(rtx_block_update(ctx, handle, size, data));
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
    RTX_FORWARD_TEXTURE(rtx_texture2d_create(ctx, width, height, channels, texels, wrap, handle));
    return RTX_OK;
}

int rtx_cubemap_create(rtx_context* ctx, int face_size, int channels, const uint8_t* const faces[6], int gen_mipmap, uint32_t* handle)
{
    if (!ctx || !handle || !faces) return fail(RTX_ERR_INVALID, "rtx_cubemap_create: null argument");
    *handle = 0;
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
        if (gen_mipmap) {
            // load_cubemap(faces, genMipmap = true): glGenerateMipmap(GL_TEXTURE_CUBE_MAP) + GL_LINEAR_MIPMAP_LINEAR (GLWrapper.cpp:307-310), so
            // texture(skybox, rd) (rt.frag:893) is trilinear. Every face gets the chain of a 2-D image of its own (the same rounded integer
            // mean); device layout: level L = 6 faces of max(1, size>>L)^2 dwords, behind level L-1 (rt_scene_dev.h DevCubemap)
            std::vector<uint32_t> chain[6];
            uint32_t off[MAX_MIPS] = {0};
            for (int f = 0; f < 6; f++) {
                chain[f].assign(host.begin() + static_cast<std::ptrdiff_t>(fsz * f), host.begin() + static_cast<std::ptrdiff_t>(fsz * (f + 1)));
                t.levels = rtpack::build_mip_chain(chain[f], face_size, face_size, off, MAX_MIPS);
            }
            for (int l = 1; l < t.levels; l++) {
                const size_t n = (l + 1 < t.levels ? off[l + 1] : chain[0].size()) - off[l];
                for (int f = 0; f < 6; f++) host.insert(host.end(), chain[f].begin() + off[l], chain[f].begin() + off[l] + static_cast<std::ptrdiff_t>(n));
            }
            // A face that failed to load (GLWrapper.cpp:296-305 skips it) leaves the texture cube-incomplete: glGenerateMipmap raises
            // GL_INVALID_OPERATION, and with a mip-mapping minification filter an incomplete texture samples (0, 0, 0, 1) on EVERY face.
            if (t.face_mask != 0x3f) t.face_mask = 0;
        }
        t.dwords = host.size();
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t.d_texels), t.dwords * 4));
        HIP_TRY(hipMemcpy(t.d_texels, host.data(), t.dwords * 4, hipMemcpyHostToDevice));
    }
    const uint32_t h = ctx->next_handle++;
    ctx->textures[h] = t;
    *handle = h;
    RTX_FORWARD_TEXTURE(rtx_cubemap_create(ctx, face_size, channels, faces, gen_mipmap, handle));
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
    RTX_FORWARD(rtx_sampler_unit(ctx, sampler_name, unit));
    return RTX_OK;
}

int rtx_bind_texture(rtx_context* ctx, int unit, uint32_t handle)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "rtx_bind_texture: no current context");
    if (unit < 0 || unit >= UNIT_COUNT) return fail(RTX_ERR_INVALID, "texture unit %d out of range", unit);
    RTX_FORWARD(rtx_bind_texture(ctx, unit, handle));
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
    RTX_FORWARD(rtx_texture_destroy(ctx, handle));
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
        case RTX_OPT_RAY_PENCILS: if (ctx->opt_pencils != (value != 0)) ctx->scene_dirty = true; ctx->opt_pencils = value != 0; break;   // masks are (re)built with the scene
        case RTX_OPT_GATHER_TARGETS: if (value < 1 || value > 3) return fail(RTX_ERR_INVALID, "RTX_OPT_GATHER_TARGETS: 1, 2 or 3"); ctx->gather_targets = value; break;
        case RTX_OPT_GATHER_RGB:
            if (ctx->gather_rgb != (value != 0) && ctx->banded && !ctx->owner) { int st = multi_sync(ctx); if (st) return st; }   // frames in flight finish as they started
            ctx->gather_rgb = value != 0;
            break;
        case RTX_OPT_BAND_LAYOUT:
            if (value < 0 || value > 2) return fail(RTX_ERR_INVALID, "RTX_OPT_BAND_LAYOUT: 0 (interleaved), 1 (contiguous) or 2 (contiguous, re-balanced)");
            if (value == 2 && ctx->per_process) return fail(RTX_ERR_INVALID, "RTX_OPT_BAND_LAYOUT 2 needs all ranks in one process (rtx_create_multi): a per-process group "
                                                                               "names its split with rtx_set_band_split");
            if (ctx->band_layout != value && ctx->banded && !ctx->owner) {
                int st = multi_sync(ctx);      // frames in flight finish under the layout they started with
                if (st) return st;
                if (value != 0 && ctx->split_rows.empty()) split_equal(ctx);
                ctx->resplit_frame = ctx->frame_no;
            }
            ctx->band_layout = value;
            return RTX_OK;                     // the root (or the rank itself) holds the layout: nothing to forward
        default: return fail(RTX_ERR_INVALID, "unknown option %d", option);
    }
    RTX_FORWARD(rtx_set_option(ctx, option, value));
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
        case RTX_OPT_RAY_PENCILS: *value = ctx->opt_pencils; break;
        case RTX_OPT_GATHER_TARGETS: *value = ctx->gather_targets; break;
        case RTX_OPT_GATHER_RGB: *value = ctx->gather_rgb; break;
        case RTX_OPT_BAND_LAYOUT: *value = ctx->band_layout; break;
        default: return fail(RTX_ERR_INVALID, "unknown option %d", option);
    }
    return RTX_OK;
}

int rtx_draw(rtx_context* ctx)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "null context");
    if (ctx->owner) return fail(RTX_ERR_INVALID, "rtx_draw on a peer of a multi-device context: draw through the root");
    int st;
    if (ctx->banded) {
        if (ctx->smaa_preset >= 0 && !(ctx->gather_targets & 2)) return fail(RTX_ERR_ORDER, "SMAA needs the RGBA8 target on the root: RTX_OPT_GATHER_TARGETS must include 2");
        st = multi_draw(ctx);
        if (st == RTX_OK && ctx->smaa_preset >= 0 && ctx->rank == 0) {   // the post-process runs on the root once the frame is assembled
            st = use_device(ctx);
            if (st == RTX_OK) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->moved[(ctx->frame_no - 1u) & 1u], 0));
            if (st == RTX_OK) st = smaa_resolve(ctx, ctx->stream);
        }
        return st;
    }
    const int band = ((ctx->height + 7) / 8) * 8;
    st = draw_impl(ctx, band, 0, 1, ctx->d_fb_f32, ctx->d_fb_u8, ctx->stream);
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
    return upload_smaa_tables(ctx, area_rg8, search_r8);
}

int rtx_smaa_default_tables(uint8_t* area_rg8, size_t area_bytes, uint8_t* search_r8, size_t search_bytes)
{
    if ((area_rg8 && area_bytes < static_cast<size_t>(rtx_smaa::AREA_BYTES)) || (search_r8 && search_bytes < static_cast<size_t>(rtx_smaa::SEARCH_BYTES)))
        return fail(RTX_ERR_INVALID, "rtx_smaa_default_tables: the area table is %d bytes (160 x 560 RG8), the search table %d (64 x 16 R8)", rtx_smaa::AREA_BYTES, rtx_smaa::SEARCH_BYTES);
    const uint8_t *a = nullptr, *s = nullptr;
    default_smaa_tables(&a, &s);
    if (area_rg8) std::memcpy(area_rg8, a, rtx_smaa::AREA_BYTES);
    if (search_r8) std::memcpy(search_r8, s, rtx_smaa::SEARCH_BYTES);
    return RTX_OK;
}

int rtx_enable_smaa(rtx_context* ctx, int preset)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "null context");
    if (preset < -1 || preset > RTX_SMAA_ULTRA) return fail(RTX_ERR_INVALID, "unknown SMAA preset %d", preset);
    if (preset < 0 || ctx->smaa_preset < 0) ctx->screen_valid = false;   // (re-)enabled: the screen is the colour target until a resolve has run
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
    if (ctx->banded && ctx->rank != 0) return fail(RTX_ERR_ORDER, "the frame is assembled on rank 0: rank %d has nothing to resolve", ctx->rank);
    int st = use_device(ctx);
    if (st) return st;
    if ((st = multi_sync(ctx)) != RTX_OK) return st;   // a multi-device root: the RGBA8 target is assembled on the transfer stream
    return smaa_resolve(ctx, ctx->stream);
}

int rtx_write_pixels(rtx_context* ctx, int format, const void* src_host, size_t src_bytes)
{
    if (!ctx || !src_host) return fail(RTX_ERR_INVALID, "rtx_write_pixels: null argument");
    if (format != RTX_RGBA8) return fail(RTX_ERR_INVALID, "rtx_write_pixels: only the RGBA8 colour target can be written");
    const size_t need = static_cast<size_t>(ctx->width) * ctx->height * 4;
    if (src_bytes < need) return fail(RTX_ERR_INVALID, "source holds %zu bytes, %zu needed", src_bytes, need);
    if (ctx->banded && ctx->rank != 0) return fail(RTX_ERR_ORDER, "the colour target lives on rank 0");
    int st = use_device(ctx);
    if (st) return st;
    if ((st = multi_sync(ctx)) != RTX_OK) return st;   // a multi-device root: the transfer stream may still be placing bands in this target
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(ctx->d_fb_u8, src_host, need, hipMemcpyHostToDevice));
    return RTX_OK;
}

int rtx_draw_rows(rtx_context* ctx, int row_first, int n_rows, void* dst_device, int format, void* stream)
{
    if (!ctx || !dst_device) return fail(RTX_ERR_INVALID, "rtx_draw_rows: null argument");
    if (ctx->banded) return fail(RTX_ERR_INVALID, "rtx_draw_rows on a multi-device context: it splits the frame itself (rtx_draw)");
    if (row_first < 0 || (row_first % 8) != 0 || n_rows < 0 || row_first + n_rows > ctx->height) return fail(RTX_ERR_INVALID, "rtx_draw_rows: rows [%d, %d) of a %d-row frame (the first must be a multiple of 8)", row_first, row_first + n_rows, ctx->height);
    // a range that ends inside the frame ends on a tile boundary, like the library's own splits (rtx_set_band_split): the derivative quads of
    // the texture LOD pair rows 2k and 2k + 1, and a range cut between them would difference its last row against a lane that is not traced
    if ((n_rows % 8) != 0 && row_first + n_rows != ctx->height) return fail(RTX_ERR_INVALID, "rtx_draw_rows: %d rows from row %d -- a range that does not end the frame must be a multiple of 8 rows", n_rows, row_first);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    if (format == RTX_RGBA32F) return draw_impl(ctx, 8, row_first / 8, 1, static_cast<float*>(dst_device), nullptr, s, n_rows);
    if (format == RTX_RGBA8) return draw_impl(ctx, 8, row_first / 8, 1, nullptr, static_cast<uint32_t*>(dst_device), s, n_rows);
    return fail(RTX_ERR_INVALID, "unknown format %d", format);
}
int rtx_draw_bands(rtx_context* ctx, int band_rows, int band_first, int band_stride, void* dst_device, int format, void* stream)
{
    if (!ctx || !dst_device) return fail(RTX_ERR_INVALID, "rtx_draw_bands: null argument");
    if (ctx->banded) return fail(RTX_ERR_INVALID, "rtx_draw_bands on a multi-device context: it splits the frame itself (rtx_draw)");
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
    for (rtx_context* p : ctx->peers) {
        if ((st = use_device(p)) != RTX_OK) return st;
        HIP_TRY(hipStreamSynchronize(p->stream));
        if ((st = wait_stream_bounded(p, p->xfer_stream, "gather (peer)")) != RTX_OK) return st;
    }
    return multi_sync(ctx);   // the root's (or this process' rank's) transfer stream
}

// which buffer a colour-target format names right now (nullptr + message on failure); shared by rtx_read_pixels / rtx_framebuffer_device
static int resolve_format(rtx_context* ctx, int format, const void** src, size_t* bytes)
{
    const size_t px = static_cast<size_t>(ctx->width) * ctx->height;
    if (ctx->banded && ctx->rank != 0) return fail(RTX_ERR_ORDER, "the frame is assembled on rank 0: rank %d holds only its own bands", ctx->rank);
    // the screen is the SMAA output once a resolve has run, else what the colour target holds (GLWrapper.cpp:195-204 vs :159-165)
    const bool screen_is_smaa = ctx->smaa_preset >= 0 && ctx->d_screen && ctx->screen_valid;
    switch (format) {
        case RTX_RGBA32F: *src = ctx->d_fb_f32; *bytes = px * 16; break;
        case RTX_RGBA8: *src = ctx->d_fb_u8; *bytes = px * 4; break;
        case RTX_SCREEN_RGBA8: *src = screen_is_smaa ? ctx->d_screen : ctx->d_fb_u8; *bytes = px * 4; break;
        case RTX_SMAA_EDGES_RG8: *src = ctx->d_edges; *bytes = px * 2; break;
        case RTX_SMAA_WEIGHTS_RGBA8: *src = ctx->d_blend; *bytes = px * 4; break;
        default: return fail(RTX_ERR_INVALID, "unknown format %d", format);
    }
    if (!*src) return fail(RTX_ERR_ORDER, "format %d needs SMAA to be enabled (rtx_enable_smaa)", format);
    if (ctx->banded) {
        const bool needs_f32 = format == RTX_RGBA32F, needs_u8 = format == RTX_RGBA8 || (format == RTX_SCREEN_RGBA8 && !screen_is_smaa);
        if ((needs_f32 && !(ctx->gather_targets & 1)) || (needs_u8 && !(ctx->gather_targets & 2)))
            return fail(RTX_ERR_ORDER, "this colour target is not gathered (RTX_OPT_GATHER_TARGETS)");
    }
    return RTX_OK;
}

int rtx_read_pixels(rtx_context* ctx, int format, void* dst_host, size_t dst_bytes)
{
    if (!ctx || !dst_host) return fail(RTX_ERR_INVALID, "rtx_read_pixels: null argument");
    int st = use_device(ctx);
    if (st) return st;
    const void* src = nullptr;
    size_t need = 0;
    if ((st = resolve_format(ctx, format, &src, &need)) != RTX_OK) return st;
    if (dst_bytes < need) return fail(RTX_ERR_INVALID, "destination holds %zu bytes, %zu needed", dst_bytes, need);
    if ((st = multi_sync(ctx)) != RTX_OK) return st;
    if (format == RTX_SMAA_EDGES_RG8 || format == RTX_SMAA_WEIGHTS_RGBA8)   // kept as bit planes / never cleared on the device: made on demand
        HIP_TRY(smaa_expand(smaa_buffers(ctx, ctx->smaa_frame - 1u), ctx->stream));   // the last resolve's planes
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(dst_host, src, need, hipMemcpyDeviceToHost));
    return RTX_OK;
}

int rtx_framebuffer_device(rtx_context* ctx, int format, void** device_ptr)
{
    if (!ctx || !device_ptr) return fail(RTX_ERR_INVALID, "null argument");
    if (format != RTX_RGBA32F && format != RTX_RGBA8 && format != RTX_SCREEN_RGBA8) return fail(RTX_ERR_INVALID, "unknown format %d", format);
    const void* src = nullptr;
    size_t bytes = 0;
    const int st = resolve_format(ctx, format, &src, &bytes);
    if (st) return st;
    *device_ptr = const_cast<void*>(src);
    return RTX_OK;
}

static int get_stats_impl(rtx_context* ctx, rtx_stats* out);
int rtx_get_stats(rtx_context* ctx, rtx_stats* out) { return get_stats_impl(ctx, out); }
// rtx_stats grows with the library (rounds 2 and 3 appended fields). A caller compiled against an older header passes ITS struct: this
// entry point never writes beyond the size the caller names (the fields are append-only, so a prefix is a valid older struct).
int rtx_get_stats_sized(rtx_context* ctx, void* out, size_t out_bytes)
{
    if (!out || out_bytes == 0) return fail(RTX_ERR_INVALID, "rtx_get_stats_sized: null argument");
    rtx_stats full;
    const int st = get_stats_impl(ctx, &full);
    if (st != RTX_OK) return st;
    std::memcpy(out, &full, out_bytes < sizeof full ? out_bytes : sizeof full);
    return RTX_OK;
}
static int get_stats_impl(rtx_context* ctx, rtx_stats* out)
{
    if (!ctx || !out) return fail(RTX_ERR_INVALID, "null argument");
    int st = use_device(ctx);
    if (st) return st;
    st = drain_events(ctx);
    if (st) return st;
    std::memset(out, 0, sizeof *out);
    out->last_draw_ms = ctx->last_ms;
    out->launches = ctx->launches;
    for (rtx_context* p : ctx->peers) {   // slowest rank's kernel; counters summed below
        if ((st = use_device(p)) != RTX_OK || (st = drain_events(p)) != RTX_OK) return st;
        if (p->last_ms > out->last_draw_ms) out->last_draw_ms = p->last_ms;
    }
    if ((st = use_device(ctx)) != RTX_OK) return st;
    if (ctx->gather_timed) {
        HIP_TRY(hipEventSynchronize(ctx->gather_stop));
        HIP_TRY(hipEventElapsedTime(&out->last_gather_ms, ctx->gather_start, ctx->gather_stop));
    }
    out->pencils = ctx->n_pencil;
    out->kernel_variant = ctx->last_variant;
    out->candidate_tables = ctx->last_tables;
    if (ctx->pencil_timed) {
        HIP_TRY(hipEventSynchronize(ctx->pencil_stop));
        HIP_TRY(hipEventElapsedTime(&out->last_pencil_build_ms, ctx->pencil_start, ctx->pencil_stop));
    }
    if (ctx->smaa_timed) {
        HIP_TRY(hipEventSynchronize(ctx->smaa_stop));
        HIP_TRY(hipEventElapsedTime(&out->last_smaa_ms, ctx->smaa_start, ctx->smaa_stop));
        std::vector<uint32_t> n(SMAA_COUNT_SET);
        HIP_TRY(hipMemcpy(n.data(), ctx->d_smaa_count + ((ctx->smaa_frame - 1u) & 1u) * SMAA_COUNT_SET, n.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (int k = 0; k < SMAA_SEGMENTS; k++) out->smaa_edge_pixels += n[static_cast<size_t>(k) * SMAA_COUNT_STRIDE];
    }
    if (ctx->opt_count) {
        unsigned long long c[4];
        HIP_TRY(hipMemcpy(c, ctx->d_counters, sizeof c, hipMemcpyDeviceToHost));
        out->rays_closest = c[0];
        out->rays_shadow = c[1];
        out->rays_shadow_cast = c[2];
        out->torus_solves = c[3];
        for (rtx_context* p : ctx->peers) {
            if ((st = use_device(p)) != RTX_OK) return st;
            HIP_TRY(hipStreamSynchronize(p->stream));
            HIP_TRY(hipMemcpy(c, p->d_counters, sizeof c, hipMemcpyDeviceToHost));
            out->rays_closest += c[0];
            out->rays_shadow += c[1];
            out->rays_shadow_cast += c[2];
            out->torus_solves += c[3];
        }
        if ((st = use_device(ctx)) != RTX_OK) return st;
    }
    return RTX_OK;
}

// Extra diagnostics (not part of the reference surface) ---------------------------------------
// Sum of the HIP-event durations of the `n` most recent draws (n <= 128), for benches that time
// K draws back to back. Returns RTX_ERR_INVALID if fewer than n draws are pending.
static int sum_recent_one(rtx_context* ctx, int n, float* sum_ms)
{
    if (n > ctx->ev_pending) return fail(RTX_ERR_INVALID, "rtx_sum_recent_draw_ms: %d draws requested, %d pending", n, ctx->ev_pending);
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

// A multi-device context reports the slowest rank's sum (the ranks trace concurrently: that is the frame's trace time).
RTX_API int rtx_sum_recent_draw_ms(rtx_context* ctx, int n, float* sum_ms)
{
    if (!ctx || !sum_ms || n <= 0) return fail(RTX_ERR_INVALID, "rtx_sum_recent_draw_ms: bad arguments");
    int st = sum_recent_one(ctx, n, sum_ms);
    if (st) return st;
    for (rtx_context* p : ctx->peers) {
        float v = 0.0f;
        if ((st = sum_recent_one(p, n, &v)) != RTX_OK) return st;
        if (v > *sum_ms) *sum_ms = v;
    }
    return use_device(ctx);
}

// The same durations one by one (most recent first), without retiring them: a bench that wants the spread of its K draws (min / median /
// max) calls this BEFORE rtx_sum_recent_draw_ms. A multi-device context reports, per draw, the slowest rank.
RTX_API int rtx_recent_draw_ms(rtx_context* ctx, int n, float* ms_each)
{
    if (!ctx || !ms_each || n <= 0) return fail(RTX_ERR_INVALID, "rtx_recent_draw_ms: bad arguments");
    for (int k = 0; k < n; k++) ms_each[k] = 0.0f;
    std::vector<rtx_context*> all{ctx};
    all.insert(all.end(), ctx->peers.begin(), ctx->peers.end());
    for (rtx_context* c : all) {
        if (n > c->ev_pending) return fail(RTX_ERR_INVALID, "rtx_recent_draw_ms: %d draws requested, %d pending", n, c->ev_pending);
        int st = use_device(c);
        if (st) return st;
        for (int k = 0; k < n; k++) {
            const int idx = (c->ev_head - 1 - k + EVENT_RING * 2) % EVENT_RING;
            HIP_TRY(hipEventSynchronize(c->ev_stop[idx]));
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, c->ev_start[idx], c->ev_stop[idx]));
            if (ms > ms_each[k]) ms_each[k] = ms;
        }
    }
    return use_device(ctx);
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
