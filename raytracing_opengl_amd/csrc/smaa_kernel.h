// smaa_kernel.h -- launch interface of the SMAA post-process kernels (smaa_kernel.hip) for the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#ifndef SMAA_COUNT_STRIDE
#define SMAA_COUNT_STRIDE 32   /* dwords between two segment counters: every counter in a 128-byte line of its own (smaa_kernel.hip) */
#endif
enum { SMAA_SEGMENTS = 64, SMAA_COUNT_SET = SMAA_SEGMENTS * SMAA_COUNT_STRIDE };

struct SmaaBuffers {
    int w, h;
    const uint32_t* color;   // RGBA8 colour target of the tracer (fboTexColor, GLWrapper.cpp:127)
    uint32_t* screen;        // RGBA8 output (what the reference draws to the default framebuffer, GLWrapper.cpp:195-204)
    uint16_t* edges;         // RG8 (fboTexEdge); ZERO outside the listed pixels at all times
    uint32_t* blend;         // RGBA8 (fboTexBlend); ZERO outside the listed pixels at all times
    uint32_t* list;          // pixel indices (y * w + x) of the current frame's edge pixels: SMAA_SEGMENTS segments of segment_capacity entries
    size_t segment_capacity;
    uint32_t* count;         // 2 x SMAA_SEGMENTS counters (counter k of set s at [s * SMAA_COUNT_SET + k * SMAA_COUNT_STRIDE]), the two sets used
                             // alternately by consecutive frames (see smaa_kernel.hip)
    uint64_t* bits;          // the edge texture again as bit planes, written densely every frame (smaa_device.h PlaneEdges) -- rows: h rows of
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
hipError_t smaa_launch(const SmaaBuffers& b, int preset, unsigned frame, hipStream_t stream);
