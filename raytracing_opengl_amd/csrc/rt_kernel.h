// rt_kernel.h -- launch interface between the C-ABI layer (rtx_capi.cpp) and the kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "rt_device.h"

struct RtLaunchParams {
    const char* scene;        // DevScene blob (device memory)
    int32_t scene_bytes;
    int32_t fb_w, fb_h;       // framebuffer size = gl_FragCoord range
    // row-band set traced by this launch: bands band_first, +band_stride, ... of band_rows rows
    // each, stored packed (rows_local rows in total) at out_*
    int32_t band_rows, band_first, band_stride, rows_local;
    int32_t xcd_remap;        // 1: XCD-aware super-tile order (see rt_kernel.hip)
    int32_t hot_row0, hot_rows;  // workgroup rows [hot_row0, hot_row0 + hot_rows) of this launch are dispatched first (0 rows = off)
    int32_t grid_x, grid_y, st_nx, st_ny;  // filled by rt_launch_trace
    int32_t ps_fence_slot;    // filled at launch: the variant's pad slot, deliberately a run-time value (rtdev::PathStore::fence)
    float* out_f32;           // RGBA32F, 16 B/pixel, or nullptr
    uint32_t* out_u8;         // RGBA8, 4 B/pixel, or nullptr
    unsigned long long* counters;  // 4 x u64 (COUNT variant) or nullptr
    const uint32_t* pencil_masks;  // ray-pencil masks of this scene (rt_launch_pencil_build) or nullptr
    rtdev::TexTable tex;
};

// ev_start / ev_stop (both or neither): receive the launch's own begin / end timestamps; ev_stop is also what to wait on for its completion
hipError_t rt_launch_trace(const RtLaunchParams& p, bool cull, bool count, bool lds, bool high_occupancy, hipStream_t stream, hipEvent_t ev_start = nullptr,
                           hipEvent_t ev_stop = nullptr);
// Fills the ray-pencil masks of the scene at d_scene (host copy of its header and pencil records: n_pencil, cells). One thread per cell.
hipError_t rt_launch_pencil_build(const char* d_scene, const rtdev::DevSceneHeader& hdr, const rtdev::DevPencil* pencils, uint32_t* d_masks, hipStream_t stream);
hipError_t rt_launch_selftest(int* d_result, hipStream_t stream);
