/* rtx.h -- C ABI of the MI355X-native tracer (librtx_hip.so).
 *
 * This is the drop-in boundary for the reference's GLSL fragment-shader path
 * (assets/shaders/rt.frag dispatched by src/GLWrapper.cpp). Each entry point replaces one
 * GLWrapper method or one GL call the reference's main loop issues; the reference file:line is
 * cited per function. Plain pointers and sizes only -- no C++/torch types -- so that C, C++,
 * Python (ctypes) or any FFI can bind it. include/rtx/GLWrapper.h is the header-only C++ shim
 * with the reference's own class/method names on top of these calls.
 *
 * Conventions
 *   - every call returns an rtx_status (0 = ok); rtx_last_error() gives the message of the
 *     last failure on the calling thread. The reference prints and exit()s instead
 *     (GLWrapper.cpp:224-227,371-375; utils.h:57-63); the C++ shim restores that behaviour.
 *   - the library never keeps caller pointers: block and texture bytes are copied during the call
 *     (same ownership rule as glBufferData / glTexImage2D).
 *   - single-threaded per context, like a GL context.
 *   - there is NO CPU fallback: rtx_create fails with RTX_ERR_DEVICE when no gfx950 device /
 *     HIP runtime is usable.
 *   - framebuffer row 0 is the BOTTOM row (gl_FragCoord origin, rt.frag:315).
 */
#ifndef RTX_H_
#define RTX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define RTX_API __attribute__((visibility("default")))
#else
#define RTX_API
#endif

typedef struct rtx_context rtx_context;

typedef enum rtx_status {
    RTX_OK = 0,
    RTX_ERR_INVALID = 1, /* bad argument */
    RTX_ERR_DEVICE = 2,  /* HIP runtime / device failure (message has the hipError) */
    RTX_ERR_ORDER = 3,   /* call made before rtx_specialize, or draw with missing blocks */
    RTX_ERR_NAME = 4,    /* unknown uniform-block or sampler name */
    RTX_ERR_HANDLE = 5   /* unknown block / texture handle */
} rtx_status;

/* Same 60 bytes as the reference's `rt_defines` (src/scene.h:7-20): eight array sizes, the
 * bounce depth and two colours. It is the tracer's specialisation key. */
typedef struct rtx_defines {
    int32_t sphere_size, plane_size, surface_size, box_size, torus_size, ring_size;
    int32_t light_point_size, light_direct_size;
    int32_t iterations;
    float ambient_color[3];
    float shadow_ambient[3];
} rtx_defines;

typedef enum rtx_format {
    RTX_RGBA32F = 0, /* 16 B/pixel, unclamped -- the parity buffer (FragColor before write-out) */
    RTX_RGBA8 = 1,   /* 4 B/pixel, clamp [0,1] + round-to-nearest -- what GL_RGBA8 fboColor holds
                        (GLWrapper.cpp:127,209-222) */
    RTX_SCREEN_RGBA8 = 2,       /* 4 B/pixel: what the reference's WINDOW shows -- the SMAA output when SMAA is enabled
                                   (GLWrapper.cpp:195-204), else the same bytes as RTX_RGBA8. rtx_read_pixels only. */
    RTX_SMAA_EDGES_RG8 = 3,     /* 2 B/pixel: fboTexEdge after the last resolve (GLWrapper.cpp:173-180); rtx_read_pixels only */
    RTX_SMAA_WEIGHTS_RGBA8 = 4  /* 4 B/pixel: fboTexBlend after the last resolve (GLWrapper.cpp:182-193); rtx_read_pixels only */
} rtx_format;

/* enum SMAA_PRESET (src/SMAA_Builder.h:9-12) */
typedef enum rtx_smaa_preset { RTX_SMAA_OFF = -1, RTX_SMAA_LOW = 0, RTX_SMAA_MEDIUM = 1, RTX_SMAA_HIGH = 2, RTX_SMAA_ULTRA = 3 } rtx_smaa_preset;

typedef enum rtx_wrap { RTX_WRAP_REPEAT = 0, RTX_WRAP_CLAMP_TO_EDGE = 1 } rtx_wrap;

typedef enum rtx_option {
    RTX_OPT_CULL = 0,        /* 1 (default): conservative bounding culls in front of the torus /
                                quadric / box / ring tests (parity-gated); 0: literal scans */
    RTX_OPT_COUNT_RAYS = 1,  /* 1: kernel also counts closest-hit and shadow rays (rtx_stats) */
    RTX_OPT_SCENE_LDS = 2,   /* 1: stage the scene tables into LDS per workgroup; 0: scalar (SMEM) loads */
    RTX_OPT_TEXTURE_LOD = 3, /* 1 (default): mip chain + trilinear + quad-derivative LOD, the reference's texture
                                state (GLWrapper.cpp:337-343, rt.frag:326-338); 0: level-0 bilinear everywhere */
    RTX_OPT_XCD_REMAP = 4,   /* 1: workgroups are dealt to the 8 XCDs in 128x32-pixel super-tiles (texture lines stay in
                                one XCD's L2); 0 (default): plain row-major order. Measured on the 4K default scene:
                                FETCH_SIZE 96.3 vs 97.5 MB, kernel 0.939 vs 0.902 ms -- no reuse to win, so it is off. */
    RTX_OPT_HOT_ROWS_FIRST = 6, /* 1 (default): in launches of at most 24 000 workgroups (frames up to about 3200 x 1800) and in
                                rtx_draw_bands launches that cover a quarter of the frame or less (band_stride >= 4) the workgroup
                                rows that show a torus -- tiles that run ~20x the median -- are dispatched first, so that they do
                                not form the tail of the launch; 0: plain row order. Same results either way. */
    RTX_OPT_GATHER_TARGETS = 7, /* multi-device contexts: which colour targets travel to the root each draw: 1 = RGBA32F only, 2 = RGBA8
                                only (what the reference's framebuffer holds: a quarter of the bytes), 3 (default) = both */
    RTX_OPT_RAY_PENCILS = 8,  /* 1 (default): scenes with a long quadric / torus table (16 .. 128 entries) get per-pencil candidate masks --
                                camera rays by direction, shadow rays by direction from a point light or by position across a directional
                                light -- built on the device whenever the scene changes; scans of such rays walk the wave's candidates
                                instead of the whole table. 0: two-level scans only. Same results. */
    RTX_OPT_BAND_LAYOUT = 9,  /* multi-device contexts: how the frame is split over the ranks. 0 (default): interleaved 8-row bands (band b -> rank
                                b mod N: sky and object rows alternate between the ranks; the root receives into landing buffers and a copy
                                kernel puts the bands in place). 1: ONE contiguous range of rows per rank (rtx_set_band_split; equal ranges
                                until it is called): the root traces its range straight into the colour targets and receives the others
                                straight into place -- no landing buffers, no placement pass. 2: as 1, and the library moves the boundaries
                                towards equal kernel times (rates of two frames back; single-process contexts only). Same pixels in every
                                layout. A per-process group (rtx_create_rank) must set the same value on every rank. */
    RTX_OPT_GATHER_RGB = 10,  /* multi-device contexts, interleaved layout: 1 (default): the RGBA32F bands travel to the root WITHOUT their alpha
                                channel -- 12 bytes per pixel instead of 16; the traced frame's alpha is the constant 1.0f (rt.frag:902) and the
                                root writes it back while it places the bands, so the assembled frame is bit-identical; 0: 16 bytes per
                                pixel. At N = 2 ... 4 the frame rate of the float target is the rate of the root's links.
                                The contiguous layouts receive in place and always move whole pixels. A per-process group (rtx_create_rank) must
                                set the same value on every rank, like RTX_OPT_BAND_LAYOUT and rtx_set_band_split: each process sizes its own
                                side of the paired ncclSend / ncclRecv from them. Since round 6 the ranks cross-check: whenever a rank's frame
                                configuration (size, rank count, layout, split, targets, this option) differs from the one last confirmed, the
                                next rtx_draw first all-gathers 16-byte digests across the ranks and fails ON EVERY RANK with RTX_ERR_INVALID,
                                naming the rank that differs, before a band travels; waits on the transfer stream are bounded by the
                                environment variable RTX_GATHER_TIMEOUT_MS (default 30 000; 0 = wait for ever) and end with RTX_ERR_DEVICE
                                instead of a hung process when a rank never takes part. */
    RTX_OPT_HIGH_OCCUPANCY = 5 /* which build of the trace kernel runs: 0 = the default one, 1 = the many-primitive one (group culls, ray
                                pencils and slab tables compiled in; its own register budget -- 7 waves/SIMD in round 1, hence the
                                name, 6 now), -1 (default) = choose by primitive count (>= 32 -> 1). Same results. */
} rtx_option;

typedef struct rtx_stats {
    float last_draw_ms;        /* HIP-event time of the last rtx_draw* kernel(s) on their stream */
    uint32_t launches;         /* kernel launches since rtx_create */
    uint64_t rays_closest;     /* valid after a draw with RTX_OPT_COUNT_RAYS = 1 */
    uint64_t rays_shadow;      /*   (reference-defined rays: calcInter / inShadow invocations) */
    uint64_t rays_shadow_cast; /*   shadow scans the kernel actually executed (dp > 0 only) */
    uint64_t torus_solves;     /*   Durand-Kerner solves actually run */
    float last_smaa_ms;        /* HIP-event time of the last SMAA resolve (all of its kernels), 0 if none ran */
    float last_gather_ms;      /* multi-device contexts: transfer + band placement of the last frame on the root (0 otherwise) */
    uint32_t smaa_edge_pixels; /* pixels with an edge in the last resolve (the sparse passes' work list) */
    float last_pencil_build_ms; /* HIP-event time of the last ray-pencil mask build (runs when the scene changed; 0: scene has none) */
    uint32_t pencils;          /* ray pencils of the current scene (RTX_OPT_RAY_PENCILS) */
    uint32_t kernel_variant;   /* which build of the trace kernel the last draw ran: 0 = default, 1 = many-primitive (RTX_OPT_HIGH_OCCUPANCY) */
    uint32_t candidate_tables; /* candidate selection in front of the last draw's long-table scans, bit set = active: 1 = group culls
                                  (>= 16 quadrics or tori), 2 = ray pencils, 4 = slab + direction tables. 0 with many quadrics / tori
                                  means the scene fell outside what the tables hold (more than 128 of a kind: the library says so once
                                  on stderr) or RTX_OPT_RAY_PENCILS / RTX_OPT_CULL is off: plain two-level scans, same results, slower */
} rtx_stats;

RTX_API const char* rtx_last_error(void);
RTX_API const char* rtx_version(void);

/* GLWrapper::GLWrapper(w,h,fullScreen) + init_window()  [GLWrapper.cpp:12-18,61-133]
 * Creates the device context and the w x h colour target on HIP device `device`.
 * The new context becomes current (see rtx_current). */
RTX_API int rtx_create(int width, int height, int device, rtx_context** out);
/* The same on N devices of one node (BASELINE north_star: "GLWrapper dispatch -> HIP launch + RCCL tile gather"). The
 * context returned is used exactly like a single-device one: blocks, textures and options are replicated to every device, rtx_draw
 * splits the frame into interleaved 8-row bands (band b -> device b mod N), every device traces its bands, and the root (device_ids[0])
 * receives them and assembles the colour targets that rtx_read_pixels / the SMAA resolve see. `gather`: how the bands reach the root --
 * RTX_GATHER_RCCL: one grouped ncclSend/ncclRecv pair per peer (librccl.so is loaded at this call; one device per rank);
 * RTX_GATHER_PEER_COPY: hipMemcpyPeerAsync by the root (no library; ranks may share a device). n_devices = 1 is a plain context
 * (except with RTX_GATHER_RCCL_LOOPBACK). Only the colour targets RTX_OPT_GATHER_TARGETS names are traced and moved. */
typedef enum rtx_gather {
    RTX_GATHER_RCCL = 0,
    RTX_GATHER_PEER_COPY = 1,
    RTX_GATHER_RCCL_LOOPBACK = 2 /* as RTX_GATHER_RCCL, and the root's OWN bands also travel through ncclSend / ncclRecv (to itself) instead of
                                    being placed directly: a diagnostic with which a box that has a single GPU executes the RCCL transport
                                    end to end (library load, communicator, grouped send/recv, stream ordering). n_devices may be 1. */
} rtx_gather;
RTX_API int rtx_create_multi(int width, int height, int n_devices, const int* device_ids, int gather, rtx_context** out);
/* The same frame split with ONE PROCESS PER GPU (the launch model of torch.distributed.run / mpirun): every process creates the context of
 * its own rank on its own device and then drives it exactly like a single-device context -- the host program is the same on every rank, so
 * blocks, textures and options are replicated by construction. rtx_draw traces the rank's interleaved 8-row bands (band b -> rank b mod N)
 * and, on the rank's transfer stream, sends them to rank 0 (ncclSend); rank 0 receives every peer's bands inside one ncclGroup and
 * assembles the colour targets, which only rank 0 can read (rtx_read_pixels on another rank is RTX_ERR_ORDER). The RCCL communicator is
 * bootstrapped from a unique id: rank 0 calls rtx_rccl_unique_id and hands the 128 bytes to the other processes by whatever means the
 * launcher offers (a torch.distributed / MPI broadcast, a file). `gather`: RTX_GATHER_RCCL or RTX_GATHER_RCCL_LOOPBACK. Collective: every
 * rank must make the call, and every rank must call rtx_draw the same number of times. rtx_get_stats reports the rank's own counters. */
#define RTX_RCCL_ID_BYTES 128
RTX_API int rtx_rccl_unique_id(uint8_t id[RTX_RCCL_ID_BYTES]);
RTX_API int rtx_create_rank(int width, int height, int device, int rank, int n_ranks, const uint8_t id[RTX_RCCL_ID_BYTES], int gather,
                            rtx_context** out);
/* Number of ranks the frame is split over (1 for a plain context) and this context's rank (0 for a plain or single-process context). */
RTX_API int rtx_device_count(rtx_context* ctx, int* n_devices);
RTX_API int rtx_rank(rtx_context* ctx, int* rank);
/* GLWrapper::~GLWrapper / stop()  [GLWrapper.cpp:25-44,143-147] */
RTX_API void rtx_destroy(rtx_context* ctx);
/* The reference's update_buffer / load_cubemap are static and act on "the current GL context";
 * the shim uses these for the same purpose. */
RTX_API rtx_context* rtx_current(void);
RTX_API int rtx_make_current(rtx_context* ctx);
RTX_API int rtx_get_size(rtx_context* ctx, int* width, int* height); /* getWidth/getHeight */

/* GLWrapper::init_shaders(rt_defines&)  [GLWrapper.cpp:232-277]
 * Fixes array sizes, bounce depth and the two colour constants. The two colours take the same
 * "%f" text round trip as the reference's shader templating (GLWrapper.cpp:246-247,279-282).
 * Must precede block/texture calls, like the reference (names are looked up in the program). */
#define RTX_MAX_ITERATIONS 256 /* larger bounce depths are refused with RTX_ERR_INVALID (the per-pixel trip bound) */
RTX_API int rtx_specialize(rtx_context* ctx, const rtx_defines* defines);

/* GLWrapper::init_buffer(ubo,name,bindingPoint,size,data)  [GLWrapper.cpp:365-379]
 * `name` is one of scene_buf, spheres_buf, planes_buf, surfaces_buf, boxes_buf, toruses_buf,
 * rings_buf, lights_point_buf, lights_direct_buf (rt.frag:155-230). size may be 0 and data NULL.
 * Unknown name -> RTX_ERR_NAME (the reference exits). */
RTX_API int rtx_block_create(rtx_context* ctx, const char* name, int binding_point, size_t size,
                             const void* data, uint32_t* handle);
/* GLWrapper::update_buffer(ubo,size,data)  [GLWrapper.cpp:381-386] -- overwrite from offset 0. */
RTX_API int rtx_block_update(rtx_context* ctx, uint32_t handle, size_t size, const void* data);

/* GLWrapper::load_texture(path,wrap) minus the file decode  [GLWrapper.cpp:319-354]:
 * 8-bit interleaved texels, 1/3/4 channels, row 0 = t 0 (stb_image order, no flip); builds the
 * full mip chain (glGenerateMipmap), trilinear min / linear mag. */
RTX_API int rtx_texture2d_create(rtx_context* ctx, int width, int height, int channels,
                                 const uint8_t* texels, int wrap, uint32_t* handle);
/* GLWrapper::load_cubemap(faces,genMipmap) minus the file decode  [GLWrapper.cpp:284-317]:
 * faces in +X,-X,+Y,-Y,+Z,-Z order; a NULL face is skipped (stays black) like a face that
 * failed to load. LINEAR, CLAMP_TO_EDGE, not seamless.
 * gen_mipmap = 0 is the reference's default (what main.cpp:137-147 passes): the sky is sampled at level 0. gen_mipmap != 0 is
 * load_cubemap(faces, true) (GLWrapper.cpp:307-310: glGenerateMipmap(GL_TEXTURE_CUBE_MAP), GL_LINEAR_MIPMAP_LINEAR): every face gets a mip
 * chain (the 2-D textures' rounded integer mean) and texture(skybox, rd) (rt.frag:893) is trilinear, its level of detail from the 2x2
 * quad's direction differences projected on the pixel's own face (cube part); needs RTX_OPT_TEXTURE_LOD = 1 (the
 * default; with 0 level 0 is sampled, as for the 2-D textures) and is not available together with RTX_OPT_SCENE_IN_LDS = 1 (rtx_draw
 * fails with RTX_ERR_INVALID). On any failure *handle is 0. */
RTX_API int rtx_cubemap_create(rtx_context* ctx, int face_size, int channels,
                               const uint8_t* const faces[6], int gen_mipmap, uint32_t* handle);
/* shader.setInt(uniformName, unit)  [GLWrapper.cpp:138,360]: sampler names skybox,
 * texture_sphere_1..4, texture_ring, texture_box (rt.frag:136-143). */
RTX_API int rtx_sampler_unit(rtx_context* ctx, const char* sampler_name, int unit);
/* glActiveTexture(GL_TEXTURE0+unit); glBindTexture(target, handle)  [main.cpp:178-187,
 * GLWrapper.cpp:139-140]. The target is the texture's own kind; handle 0 clears both the 2-D and
 * the cube binding of the unit. */
RTX_API int rtx_bind_texture(rtx_context* ctx, int unit, uint32_t handle);
RTX_API int rtx_texture_destroy(rtx_context* ctx, uint32_t handle);

RTX_API int rtx_set_option(rtx_context* ctx, int option, int value);
RTX_API int rtx_get_option(rtx_context* ctx, int option, int* value);

/* GLWrapper::draw()  [GLWrapper.cpp:155-165]: trace one frame from the CURRENT block and texture
 * contents into the context's own colour target (both formats are written). Asynchronous on the
 * context's stream; rtx_read_pixels / rtx_finish synchronise. */
RTX_API int rtx_draw(rtx_context* ctx);
/* Multi-GPU / external-target form of draw(): trace only the row bands
 * band_first, band_first+band_stride, ... (each `band_rows` rows, band b covers rows
 * [b*band_rows, min((b+1)*band_rows, H)) ) and store them PACKED, band after band, into `dst`
 * (a device pointer to >= rows*W pixels of `format`). `stream` is a hipStream_t (NULL = the
 * context's stream). band_rows must be a multiple of 8. */
RTX_API int rtx_draw_bands(rtx_context* ctx, int band_rows, int band_first, int band_stride,
                           void* dst_device, int format, void* stream);
/* The same for ONE contiguous range of rows -- a rank's share in the contiguous band layout (RTX_OPT_BAND_LAYOUT 1): rows
 * [row_first, row_first + n_rows), row_first a multiple of 8 and n_rows a multiple of 8 unless the range ends the frame (the texture
 * LOD's 2 x 2 derivative quads must not straddle two ranges; RTX_ERR_INVALID otherwise), stored from the start of `dst`. */
RTX_API int rtx_draw_rows(rtx_context* ctx, int row_first, int n_rows, void* dst_device, int format, void* stream);
RTX_API int rtx_finish(rtx_context* ctx);

/* ---- SMAA post-process: the three passes GLWrapper::draw runs after the tracer (GLWrapper.cpp:173-204) ----
 * GLWrapper::enable_SMAA(preset)  [GLWrapper.cpp:149-153]. With a preset >= 0 every rtx_draw is followed by the resolve of the
 * RGBA8 colour target into the screen buffer (RTX_SCREEN_RGBA8); RTX_SMAA_OFF switches it off again. Unlike the reference
 * (which must be told before init_window) this may be called at any time. */
RTX_API int rtx_enable_smaa(rtx_context* ctx, int preset);
/* SMAA_Builder::load_area_texture / load_search_texture  [SMAA_Builder.h:52-83]: the two look-up tables as the caller's
 * bytes -- area: 160 x 560 texels of RG8, search: 64 x 16 texels of R8, row 0 first (the arrays of the reference's AreaTex.h /
 * SearchTex.h have exactly this form). OPTIONAL: without this call the library uses its own tables, computed from their published
 * construction (include/rtx/smaa_tables.h) and byte-identical to the reference's two arrays -- enable_SMAA then works like the
 * reference's without any table on the caller's include path. */
RTX_API int rtx_smaa_set_tables(rtx_context* ctx, const uint8_t* area_rg8, int area_w, int area_h,
                                const uint8_t* search_r8, int search_w, int search_h);
/* Those own tables as bytes (either pointer may be NULL; 160*560*2 and 64*16 bytes). Host-only: needs no context and no device. */
RTX_API int rtx_smaa_default_tables(uint8_t* area_rg8, size_t area_bytes, uint8_t* search_r8, size_t search_bytes);
/* The post-process alone, on whatever the RGBA8 colour target holds (GLWrapper.cpp:173-204 without :155-165). */
RTX_API int rtx_smaa_resolve(rtx_context* ctx);
/* glTexSubImage2D on fboTexColor: replace the RGBA8 colour target by W*H*4 caller bytes, row 0 = bottom row (tests and
 * post-process-only use; the tracer overwrites it at the next rtx_draw). */
RTX_API int rtx_write_pixels(rtx_context* ctx, int format, const void* src_host, size_t src_bytes);
/* glReadPixels equivalent for tests/tools: copy the colour target to host memory. */
RTX_API int rtx_read_pixels(rtx_context* ctx, int format, void* dst_host, size_t dst_bytes);
/* Device pointer of the colour target (W*H pixels of `format`), for zero-copy consumers. */
RTX_API int rtx_framebuffer_device(rtx_context* ctx, int format, void** device_ptr);
/* Contiguous bands (RTX_OPT_BAND_LAYOUT 1): rank r traces rows_per_rank[r] rows, in rank order from row 0 (the bottom row). Every count
 * but the last non-zero one must be a multiple of 8 (the kernel's tile height) and together they must cover the frame. Switches layout 0
 * to 1. In a per-process group every rank must make the same call with the same numbers (the launcher has them from rtx_get_rank_draw_ms
 * of the warm-up frames, say). rtx_get_band_split reports the split in use (any layout: rows per rank). */
RTX_API int rtx_set_band_split(rtx_context* ctx, const int* rows_per_rank, int n_ranks);
RTX_API int rtx_get_band_split(rtx_context* ctx, int* rows_per_rank, int n_ranks);
/* Kernel time (HIP events, ms) of every rank's last finished trace launch: all ranks of a single-process context; in a per-process
 * group only the caller's own entry (the others are -1). Waits for the launches issued so far. */
RTX_API int rtx_get_rank_draw_ms(rtx_context* ctx, float* ms_per_rank, int n_ranks);
RTX_API int rtx_get_stats(rtx_context* ctx, rtx_stats* out);
/* rtx_stats is append-only and has grown (last_smaa_ms and everything behind it came after round 1): rtx_get_stats writes sizeof(rtx_stats)
 * of THIS header. A caller built against an older header -- or one that wants to stay binary compatible with newer libraries -- passes the
 * size of its own struct here and gets that prefix, never a write beyond it. */
RTX_API int rtx_get_stats_sized(rtx_context* ctx, void* out, size_t out_bytes);

/* ---- diagnostics without a reference counterpart ---- */
/* Sum of the HIP-event durations (ms) of the n most recent draws (n <= 128): the kernel time a
 * bench needs when it enqueues K draws back to back. A context should be driven from ONE stream. */
RTX_API int rtx_sum_recent_draw_ms(rtx_context* ctx, int n, float* sum_ms);
/* The same durations one by one, ms_each[0] = the most recent draw (n <= 128); does not retire them (call it before rtx_sum_recent_draw_ms). */
RTX_API int rtx_recent_draw_ms(rtx_context* ctx, int n, float* ms_each);
/* Device-side exhaustive check of the divide-free byte->float conversion used by the samplers
 * (must report 0 mismatches against byte/255.0f). */
RTX_API int rtx_selftest(rtx_context* ctx, int* mismatches);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H_ */
