// bands_kernel.hip -- placement of one rank's packed row bands into the assembled frame (multi-device contexts, rtx_create_multi).
//
// A rank traces the bands band_first, band_first + band_stride, ... of band_rows rows each and stores them PACKED, band after band
// (rtx_draw_bands). After the gather the root holds every rank's packed rows; this copy kernel moves them to their rows of the
// frame: 16 bytes per lane, rows contiguous on both sides, so every wave moves whole 1 KiB row segments.
#include "bands_kernel.h"

#include "band_math.h"

namespace {

template <typename Unit>
__global__ __launch_bounds__(256) void unpack_kernel(const Unit* __restrict__ packed, Unit* __restrict__ frame, int units_per_row, int fb_h,
                                                     int band_rows, int band_first, int band_stride, int rows_local)
{
    const size_t n = (size_t)rows_local * (size_t)units_per_row;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int lr = (int)(i / (size_t)units_per_row), u = (int)(i - (size_t)lr * (size_t)units_per_row);
        const int y = rtbands::frame_row_of_packed(lr, band_rows, band_first, band_stride);
        if (y < fb_h) frame[(size_t)y * (size_t)units_per_row + (size_t)u] = packed[i];
    }
}

// The RGBA32F target without its alpha channel (round 4). The tracer writes vec4(colour, 1.0) (rt.frag:902, rt_device.h trace_pixel), so a
// rank's float bands travel as 12 bytes per pixel and the root writes the 1.0f back while it places them: a quarter less on the link that
// bounds the frame rate at N = 2 ... 4. One pixel per thread: a wave reads 1 KiB and writes 768 contiguous bytes (pack),
// or the reverse (place).
__global__ __launch_bounds__(256) void pack_rgb_kernel(const float4* __restrict__ rgba, float* __restrict__ rgb, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = rgba[i];
        rgb[3 * i] = v.x;
        rgb[3 * i + 1] = v.y;
        rgb[3 * i + 2] = v.z;
    }
}
__global__ __launch_bounds__(256) void unpack_rgb_kernel(const float* __restrict__ rgb, float4* __restrict__ frame, int fb_w, int fb_h, int band_rows,
                                                         int band_first, int band_stride, int rows_local)
{
    const size_t n = (size_t)rows_local * (size_t)fb_w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int lr = (int)(i / (size_t)fb_w), x = (int)(i - (size_t)lr * (size_t)fb_w);
        const int y = rtbands::frame_row_of_packed(lr, band_rows, band_first, band_stride);
        if (y < fb_h) frame[(size_t)y * (size_t)fb_w + (size_t)x] = make_float4(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], 1.0f);
    }
}

}  // namespace

hipError_t bands_pack_rgb(const void* rgba32f, void* rgb32f, size_t n_pixels, hipStream_t stream)
{
    if (n_pixels == 0) return hipSuccess;
    const int blocks = (int)((n_pixels + 255) / 256 < 4096 ? (n_pixels + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_rgb_kernel, dim3(blocks), dim3(256), 0, stream, static_cast<const float4*>(rgba32f), static_cast<float*>(rgb32f), n_pixels);
    return hipGetLastError();
}
hipError_t bands_unpack_rgb(const void* rgb32f, void* frame_rgba32f, int fb_w, int fb_h, int band_rows, int band_first, int band_stride, int rows_local, hipStream_t stream)
{
    if (rows_local <= 0) return hipSuccess;
    const size_t n = (size_t)rows_local * (size_t)fb_w;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(unpack_rgb_kernel, dim3(blocks), dim3(256), 0, stream, static_cast<const float*>(rgb32f), static_cast<float4*>(frame_rgba32f), fb_w, fb_h, band_rows,
                       band_first, band_stride, rows_local);
    return hipGetLastError();
}

hipError_t bands_unpack(const void* packed, void* frame, int fb_w, int fb_h, int px_bytes, int band_rows, int band_first, int band_stride, int rows_local,
                        hipStream_t stream)
{
    if (rows_local <= 0) return hipSuccess;
    const size_t row_bytes = (size_t)fb_w * (size_t)px_bytes;
    if ((row_bytes & 15) == 0) {
        const int upr = (int)(row_bytes / 16);
        const size_t n = (size_t)rows_local * upr;
        const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(unpack_kernel<uint4>, dim3(blocks), dim3(256), 0, stream, static_cast<const uint4*>(packed), static_cast<uint4*>(frame), upr, fb_h,
                           band_rows, band_first, band_stride, rows_local);
    } else {
        const int upr = (int)(row_bytes / 4);
        const size_t n = (size_t)rows_local * upr;
        const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(unpack_kernel<uint32_t>, dim3(blocks), dim3(256), 0, stream, static_cast<const uint32_t*>(packed), static_cast<uint32_t*>(frame), upr, fb_h,
                           band_rows, band_first, band_stride, rows_local);
    }
    return hipGetLastError();
}
