// bands_harness.cpp -- TEST INFRASTRUCTURE: the multi-device frame's bookkeeping (raytracing_opengl_amd/csrc/band_math.h -- the functions
// rtx_capi.cpp multi_draw / multi_draw_contiguous / rebalance and the placement kernels of bands_kernel.hip call) run on the host against a
// FAKE transport: every "rank" fills its packed buffer with a function of (frame row, column), exactly the bytes band_math.h says travel are
// memcpy'd to the "root", and the root places them with the same index arithmetic the kernels use. The CPU suite thereby covers what
// `bench.py --gpus N` runs apart from the HIP / RCCL calls themselves. Not a product path.
#include <cstdint>
#include <cstring>
#include <vector>

#include "band_math.h"

namespace {
// what the tracer would write at pixel (x, y): four floats (alpha 1.0f, like rt.frag:902) or one RGBA8 dword
inline float texel_f(int y, int x, int c) { return c == 3 ? 1.0f : static_cast<float>((y * 7919 + x * 104729 + c * 31) % 100003) * 1e-3f; }
inline uint32_t texel_u(int y, int x) { return static_cast<uint32_t>(y) * 2654435761u ^ static_cast<uint32_t>(x) * 40503u; }
void trace_row(unsigned char* dst, int y, int width, int target)
{
    for (int x = 0; x < width; x++) {
        if (target == 0) { float v[4] = {texel_f(y, x, 0), texel_f(y, x, 1), texel_f(y, x, 2), texel_f(y, x, 3)}; std::memcpy(dst + static_cast<size_t>(x) * 16, v, 16); }
        else { const uint32_t v = texel_u(y, x); std::memcpy(dst + static_cast<size_t>(x) * 4, &v, 4); }
    }
}
long long check_frame(const std::vector<unsigned char>& frame, const std::vector<int>& written, int height, int width, int target)
{
    long long bad = 0;
    const size_t px = rtbands::target_bytes(target);
    std::vector<unsigned char> want(static_cast<size_t>(width) * px);
    for (int y = 0; y < height; y++) {
        if (written[y] != 1) { bad += width; continue; }      // a row nobody placed, or placed twice
        trace_row(want.data(), y, width, target);
        const unsigned char* got = frame.data() + static_cast<size_t>(y) * width * px;
        for (size_t i = 0; i < want.size(); i++) bad += got[i] != want[i];
    }
    return bad;
}
}  // namespace

extern "C" {

// Interleaved layout (RTX_OPT_BAND_LAYOUT 0). Returns the number of wrong bytes / rows in the assembled frame (0 = correct), -1 if the
// ranks' row counts do not add up to the frame.
long long bands_sim_interleaved(int height, int width, int band_rows, int n_ranks, int target, int rgb, int frames)
{
    const size_t px = rtbands::target_bytes(target);
    const bool strip = rgb != 0 && target == 0;
    long long total = 0, bad = 0;
    for (int r = 0; r < n_ranks; r++) total += rtbands::rows_interleaved(height, band_rows, n_ranks, r);
    if (total != height) return -1;
    // two buffer sets per rank, as the library keeps them: frame k uses set k & 1
    std::vector<std::vector<unsigned char>> packed[2], stage[2];
    for (int q = 0; q < 2; q++) { packed[q].resize(n_ranks); stage[q].resize(n_ranks); }
    for (int k = 0; k < frames; k++) {
        const int par = rtbands::buffer_set(static_cast<unsigned long long>(k));
        std::vector<unsigned char> frame(static_cast<size_t>(height) * width * px, 0xee);
        std::vector<int> written(height, 0);
        for (int r = 0; r < n_ranks; r++) {                  // every rank traces its bands packed back to back (rtx_draw_bands) ...
            const int rows = rtbands::rows_interleaved(height, band_rows, n_ranks, r);
            packed[par][r].assign(static_cast<size_t>(rows) * width * px, 0xcd);
            for (int lr = 0; lr < rows; lr++) trace_row(packed[par][r].data() + static_cast<size_t>(lr) * width * px, rtbands::frame_row_of_packed(lr, band_rows, r, n_ranks), width, target);
            // ... ships what band_math.h says travels (12 bytes per pixel when the float target leaves its alpha behind) ...
            const size_t nbytes = rtbands::bytes_moved(width, static_cast<size_t>(rows), target, strip);
            stage[par][r].assign(nbytes, 0xab);
            if (strip) {
                for (size_t i = 0; i < static_cast<size_t>(rows) * width; i++) std::memcpy(stage[par][r].data() + i * 12, packed[par][r].data() + i * 16, 12);
            } else {
                if (nbytes != packed[par][r].size()) return -2;
                std::memcpy(stage[par][r].data(), packed[par][r].data(), nbytes);
            }
        }
        for (int r = 0; r < n_ranks; r++) {                  // ... and the root puts every rank's rows in their place (bands_unpack / bands_unpack_rgb)
            const int rows = rtbands::rows_interleaved(height, band_rows, n_ranks, r);
            for (int lr = 0; lr < rows; lr++) {
                const int y = rtbands::frame_row_of_packed(lr, band_rows, r, n_ranks);
                if (y < 0 || y >= height) { bad++; continue; }
                written[y]++;
                unsigned char* dst = frame.data() + static_cast<size_t>(y) * width * px;
                if (strip) {
                    const float one = 1.0f;
                    for (int x = 0; x < width; x++) {
                        std::memcpy(dst + static_cast<size_t>(x) * 16, stage[par][r].data() + (static_cast<size_t>(lr) * width + x) * 12, 12);
                        std::memcpy(dst + static_cast<size_t>(x) * 16 + 12, &one, 4);
                    }
                } else {
                    std::memcpy(dst, stage[par][r].data() + static_cast<size_t>(lr) * width * px, static_cast<size_t>(width) * px);
                }
            }
        }
        bad += check_frame(frame, written, height, width, target);
    }
    return bad;
}

// Contiguous layout (RTX_OPT_BAND_LAYOUT 1 / 2): rows = a caller's split (checked like rtx_set_band_split checks it) or NULL for the equal
// split; ms (optional, n_ranks values) = kernel times to re-balance with before the frame is assembled (layout 2). rows_used receives the split
// in use. Returns wrong bytes / rows (0 = correct), -10 - code for a split that split_check rejects.
long long bands_sim_contiguous(int height, int width, int n_ranks, const int* rows_in, const double* ms, int target, int* rows_used)
{
    const size_t px = rtbands::target_bytes(target);
    std::vector<int> rows, start;
    if (rows_in) {
        int badr = 0; long long tot = 0;
        const int code = rtbands::split_check(height, rows_in, n_ranks, n_ranks, &badr, &tot);
        if (code) return -10 - code;
        rows.assign(rows_in, rows_in + n_ranks);
        rtbands::starts_of(rows, start);
    } else {
        rtbands::split_equal(height, n_ranks, rows, start);
    }
    if (ms) {
        std::vector<int> r2, s2;
        if (rtbands::rebalance(height, rows, std::vector<double>(ms, ms + n_ranks), r2, s2)) { rows.swap(r2); start.swap(s2); }
    }
    std::vector<unsigned char> frame(static_cast<size_t>(height) * width * px, 0xee);
    std::vector<int> written(height, 0);
    for (int r = 0; r < n_ranks; r++) {
        if (rows_used) rows_used[r] = rows[r];
        if (rows[r] < 0 || start[r] < 0 || start[r] + rows[r] > height) return -3;
        if (rows[r] > 0 && (start[r] % rtbands::TILE_ROWS) != 0) return -4;      // a range starts on a tile boundary (draw_impl: band index = start / 8)
        // the rank traces its range into a packed buffer, the transport moves rows * W * px bytes, the root receives them IN PLACE
        std::vector<unsigned char> packed(static_cast<size_t>(rows[r]) * width * px);
        for (int lr = 0; lr < rows[r]; lr++) trace_row(packed.data() + static_cast<size_t>(lr) * width * px, start[r] + lr, width, target);
        const size_t nbytes = rtbands::bytes_moved(width, static_cast<size_t>(rows[r]), target, false);
        if (nbytes != packed.size()) return -2;
        if (nbytes) std::memcpy(frame.data() + static_cast<size_t>(start[r]) * width * px, packed.data(), nbytes);
        for (int lr = 0; lr < rows[r]; lr++) written[start[r] + lr]++;
    }
    return check_frame(frame, written, height, width, target);
}

int bands_rebalance(int height, int n_ranks, const int* rows_now, const double* ms, int* rows_out, int* start_out)
{
    std::vector<int> r, s;
    if (!rtbands::rebalance(height, std::vector<int>(rows_now, rows_now + n_ranks), std::vector<double>(ms, ms + n_ranks), r, s)) return 0;
    for (int k = 0; k < n_ranks; k++) { rows_out[k] = r[k]; start_out[k] = s[k]; }
    return 1;
}
// ---- first contact (band_math.h FrameConfig / config_digest / config_first_mismatch / bounded_wait; rtx_capi.cpp config_handshake) ----
// cfg = n_ranks rows of 8 ints (width, height, n_ranks, band_rows, band_layout, gather_targets, gather_rgb, loopback) + split rows per rank
// (n_split ints each, used when band_layout != 0). The fake transport is an all-gather: every rank contributes its 16-byte digest and sees
// all of them; the return value is what EVERY rank of the product concludes (0 = agreed, 1 + r = rank r differs from rank 0). present[r] == 0:
// that rank never calls, the collective cannot complete -- the others' wait runs on a fake clock (one tick per poll) against
// `timeout_ticks` and the function returns -100 (the product's RTX_ERR_DEVICE: "did not finish within ...").
int bands_sim_handshake(int n_ranks, const int* cfg, const int* split, int n_split, const int* present, int timeout_ticks)
{
    std::vector<rtbands::ConfigDigest> digests(static_cast<size_t>(n_ranks));
    bool all_present = true;
    for (int r = 0; r < n_ranks; r++) {
        rtbands::FrameConfig fc;
        const int* c = cfg + 8 * r;
        fc.width = c[0]; fc.height = c[1]; fc.n_ranks = c[2]; fc.band_rows = c[3]; fc.band_layout = c[4]; fc.gather_targets = c[5]; fc.gather_rgb = c[6]; fc.loopback = c[7];
        if (split && n_split > 0) fc.split_rows.assign(split + static_cast<size_t>(r) * n_split, split + static_cast<size_t>(r + 1) * n_split);
        digests[r] = rtbands::config_digest(fc);
        all_present = all_present && present[r] != 0;
    }
    double clock = 0.0;
    const bool done = rtbands::bounded_wait([&] { return all_present; }, [&] { return clock; }, [&] { clock += 1.0; }, static_cast<double>(timeout_ticks));
    if (!done) return -100;
    const int wrong = rtbands::config_first_mismatch(digests);
    return wrong < 0 ? 0 : 1 + wrong;
}
int bands_rows_interleaved(int height, int band_rows, int n_ranks, int rank) { return rtbands::rows_interleaved(height, band_rows, n_ranks, rank); }
int bands_split_check(int height, const int* rows, int n, int n_ranks) { return rtbands::split_check(height, rows, n, n_ranks, nullptr, nullptr); }

}  // extern "C"
