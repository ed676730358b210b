// smaa_kernel.h -- launch interface of the SMAA post-process kernels (smaa_kernel.hip) for the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#ifndef SMAA_COUNT_STRIDE
#define SMAA_COUNT_STRIDE 32   /* dwords between two segment counters: every counter in a 128-byte line of its own (smaa_kernel.hip) */
#endif
#ifndef SMAA_SEGMENTS_N
#define SMAA_SEGMENTS_N 64     /* list segments (64 or 256): a strip appends to segment (strip index mod SMAA_SEGMENTS_N) */
#endif
enum { SMAA_SEGMENTS = SMAA_SEGMENTS_N, SMAA_COUNT_SET = SMAA_SEGMENTS * SMAA_COUNT_STRIDE };

struct SmaaBuffers {
    int w, h;
    const uint32_t* color;   // RGBA8 colour target of the tracer (fboTexColor, GLWrapper.cpp:127)
    uint32_t* screen;        // RGBA8 output (what the reference draws to the default framebuffer, GLWrapper.cpp:195-204)
    uint16_t* edges;         // RG8 (fboTexEdge): the dense kernel writes a lane's four texels of a row wherever this frame OR the previous one has
                             // an edge among them -- new edges in, stale ones out, no clearing pass
    uint32_t* blend;         // RGBA8 (fboTexBlend): the weight kernel writes the listed pixels' texels; texels of earlier frames' edge pixels stay
                             // behind and are never read (pass 3 looks at a texel only where `bits` has an edge pixel); smaa_expand zeroes them
    uint32_t* list;          // pixel indices (y * w + x) of the current frame's edge pixels: SMAA_SEGMENTS segments of segment_capacity entries
    size_t segment_capacity;
    uint32_t* count;         // 2 x SMAA_SEGMENTS counters (counter k of set s at [s * SMAA_COUNT_SET + k * SMAA_COUNT_STRIDE]), the two sets used
                             // alternately by consecutive frames (see smaa_kernel.hip)
    const uint64_t* bits_prev;   // the row plane the PREVIOUS resolve wrote (all zero before the first): what the RG8 texture still holds
    uint64_t* bits;          // the edge texture again as bit planes, written densely every frame (smaa_device.h SearchPlanes) -- rows: h rows of
                             // plane_words(w) 64-bit words, 32 pixels of a row per word, bit 2k = red, bit 2k + 1 = green of pixel k;
    uint16_t* cbits;         // columns: ((h + 7) / 8) x w 16-bit words, 8 pixels of a COLUMN per word
    const uint16_t* area;    // 160 x 560 RG8
    const uint8_t* search;   // 64 x 16 R8
};

// One SMAA resolve of `b.color` into `b.screen` (asynchronous on `stream`). `frame` is the caller's running count of resolves on
// these buffers (selects the counter). preset: 0 LOW .. 3 ULTRA.
// entries one list segment must hold for a w x h frame (every pixel of the strips that append to it)
size_t smaa_segment_capacity(int w, int h);
size_t smaa_plane_bytes(int w, int h);        // size of the row bit plane
size_t smaa_col_plane_bytes(int w, int h);    // size of the column bit plane
// ev_start / ev_stop (optional): the begin timestamp of the resolve's first kernel and the end timestamp of its last one land in them
// (hipExtLaunchKernel: no marker packets between the tracer and the resolve); ev_stop is also what to wait on for the resolve.
hipError_t smaa_launch(const SmaaBuffers& b, int preset, unsigned frame, hipStream_t stream, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// For read-backs of RTX_SMAA_EDGES_RG8 / RTX_SMAA_WEIGHTS_RGBA8: fills `edges` from the bit plane and zeroes the weight texels of pixels without an edge.
hipError_t smaa_expand(const SmaaBuffers& b, hipStream_t stream);
