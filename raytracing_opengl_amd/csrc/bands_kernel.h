// bands_kernel.h -- see bands_kernel.hip
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

hipError_t bands_unpack(const void* packed, void* frame, int fb_w, int fb_h, int px_bytes, int band_rows, int band_first, int band_stride, int rows_local,
                        hipStream_t stream);
// RGBA32F <-> RGB32F (12 bytes per pixel on the link; alpha of the traced frame is the constant 1.0f): pack a rank's packed bands, and place
// bands that arrived without alpha into the frame, writing it back.
hipError_t bands_pack_rgb(const void* rgba32f, void* rgb32f, size_t n_pixels, hipStream_t stream);
hipError_t bands_unpack_rgb(const void* rgb32f, void* frame_rgba32f, int fb_w, int fb_h, int band_rows, int band_first, int band_stride, int rows_local, hipStream_t stream);
