// bands_kernel.h -- see bands_kernel.hip
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

hipError_t bands_unpack(const void* packed, void* frame, int fb_w, int fb_h, int px_bytes, int band_rows, int band_first, int band_stride, int rows_local,
                        hipStream_t stream);
