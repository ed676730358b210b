// band_math.h -- the bookkeeping of a frame split over N ranks (row bands + one gather), as pure functions: which rows a rank
// traces, where a packed row lands in the assembled frame, how many bytes travel, how a measured imbalance moves the split. No HIP, no
// context: rtx_capi.cpp (multi_draw, multi_draw_contiguous, rtx_set_band_split, rebalance) and bands_kernel.hip (the placement kernels) call
// these, and tests/host_harness/bands_harness.cpp runs the same functions against a fake transport on the host, so the CPU suite covers the
// index arithmetic that `bench.py --gpus N` depends on. The reference has no counterpart (single GPU); the layouts are
// include/rtx.h RTX_OPT_BAND_LAYOUT 0 (interleaved bands of band_rows rows, band b -> rank b % N, packed back to back per rank) and 1 / 2
// (one contiguous range per rank, multiples of 8 rows).
#pragma once

#include <cstddef>
#include <vector>

#if defined(__HIPCC__)
#define RTB_HD __host__ __device__
#else
#define RTB_HD
#endif

namespace rtbands {

enum { TILE_ROWS = 8 };   // the kernel's wave tile is 8 rows high (rt_kernel.hip): a contiguous range starts on a multiple of it

// ---- interleaved layout ------------------------------------------------------------------------------------------------------------
// Rank `band_first` of `band_stride` ranks traces bands band_first, band_first + band_stride, ... and stores their rows packed, band after
// band. Row `lr` of that packed buffer is frame row ... (>= fb_h for the padding rows of a last, short band: the caller skips those).
RTB_HD inline int frame_row_of_packed(int lr, int band_rows, int band_first, int band_stride)
{
    const int j = lr / band_rows;
    return (band_first + j * band_stride) * band_rows + (lr - j * band_rows);
}
// rows of the frame that rank `rank` of `n_ranks` traces
inline int rows_interleaved(int height, int band_rows, int n_ranks, int rank)
{
    const int n_bands = (height + band_rows - 1) / band_rows;
    int rows = 0;
    for (int b = rank; b < n_bands; b += n_ranks) {
        const int y0 = b * band_rows, y1 = y0 + band_rows < height ? y0 + band_rows : height;
        rows += y1 - y0;
    }
    return rows;
}

// ---- contiguous layout -------------------------------------------------------------------------------------------------------------
// units of TILE_ROWS rows dealt out evenly, the first (units % N) ranks one more; the last non-empty range takes a short last unit.
// A frame with fewer units than ranks leaves the surplus ranks without rows.
inline void split_equal(int height, int n_ranks, std::vector<int>& rows, std::vector<int>& start)
{
    const int units = (height + TILE_ROWS - 1) / TILE_ROWS;
    rows.assign(n_ranks, 0);
    start.assign(n_ranks, 0);
    int y = 0;
    for (int r = 0; r < n_ranks; r++) {
        const int u = units / n_ranks + (r < units % n_ranks ? 1 : 0);
        const int n = (y + u * TILE_ROWS <= height) ? u * TILE_ROWS : (height - y > 0 ? height - y : 0);
        start[r] = y;
        rows[r] = n;
        y += n;
    }
}
// A caller's split (rtx_set_band_split): 0 = valid; 1 = wrong number of counts; 2 = a negative count (*bad = rank); 3 = a range that does
// not end the frame is not a multiple of TILE_ROWS (*bad = rank); 4 = the ranges do not cover the frame exactly (*total = what they cover)
inline int split_check(int height, const int* rows, int n, int n_ranks, int* bad, long long* total_out)
{
    if (!rows || n != n_ranks) return 1;
    long long total = 0;
    for (int r = 0; r < n_ranks; r++) {
        if (rows[r] < 0) { if (bad) *bad = r; return 2; }
        total += rows[r];
        if (total < height && (rows[r] % TILE_ROWS) != 0) { if (bad) *bad = r; return 3; }
    }
    if (total_out) *total_out = total;
    return total == height ? 0 : 4;
}
inline void starts_of(const std::vector<int>& rows, std::vector<int>& start)
{
    start.assign(rows.size(), 0);
    for (size_t r = 1; r < rows.size(); r++) start[r] = start[r - 1] + rows[r - 1];
}
// Layout 2: move the boundaries towards equal kernel times. ms[r] = the kernel time rank r measured with rows_now[r] rows (<= 0 or no rows:
// no measurement -- such a rank is taken to be as fast as the measured ones on average). rate_r = rows_r / ms_r; the new share is
// proportional to the rate, moved HALF the way, in units of TILE_ROWS rows, at least one unit per rank. Returns false (and leaves the
// outputs alone) when there is nothing to do: fewer units than ranks ("at least one unit each" cannot be met -- the rounding
// loop of round 4's version never ended for height <= 8 (N - 1)), no measurement at all, or times within 4 % of each other.
inline bool rebalance(int height, const std::vector<int>& rows_now, const std::vector<double>& ms, std::vector<int>& rows_out, std::vector<int>& start_out)
{
    const int N = static_cast<int>(rows_now.size());
    const int units = (height + TILE_ROWS - 1) / TILE_ROWS;
    if (N < 2 || static_cast<int>(ms.size()) != N || units < N) return false;
    std::vector<double> rate(N, 0.0);
    double sum = 0.0, tmin = 1e300, tmax = 0.0;
    int measured = 0, starved = 0;
    for (int r = 0; r < N; r++) {
        if (rows_now[r] <= 0 || !(ms[r] > 0.0)) { starved++; continue; }
        rate[r] = rows_now[r] / ms[r];
        sum += rate[r];
        measured++;
        tmin = ms[r] < tmin ? ms[r] : tmin;
        tmax = ms[r] > tmax ? ms[r] : tmax;
    }
    if (measured == 0 || !(sum > 0.0)) return false;
    if (starved > 0) {
        const double mean = sum / measured;
        for (int r = 0; r < N; r++)
            if (rate[r] == 0.0) { rate[r] = mean; sum += mean; }
    } else if (tmax <= 1.04 * tmin) {
        return false;                                       // balanced within the noise of the timers
    }
    std::vector<int> u(N);
    int used = 0;
    for (int r = 0; r < N; r++) {
        const double want = units * rate[r] / sum, have = rows_now[r] / static_cast<double>(TILE_ROWS);
        u[r] = static_cast<int>(have + 0.5 * (want - have) + 0.5);
        if (u[r] < 1) u[r] = 1;
        used += u[r];
    }
    // rounding: hand the difference round, one unit at a time. units >= N and every u[r] >= 1, so a surplus always finds a rank with more than
    // one unit within one round; the bound is a backstop, not a path.
    for (int r = 0, guard = 0; used != units && guard < 2 * N * (units + N); r = (r + 1) % N, guard++) {
        if (used < units) { u[r]++; used++; }
        else if (u[r] > 1) { u[r]--; used--; }
    }
    if (used != units) return false;
    rows_out.assign(N, 0);
    start_out.assign(N, 0);
    int y = 0;
    for (int r = 0; r < N; r++) {
        start_out[r] = y;
        rows_out[r] = (y + u[r] * TILE_ROWS <= height) ? u[r] * TILE_ROWS : height - y;
        y += rows_out[r];
    }
    return true;
}

// ---- what travels --------------------------------------------------------------------------------------------------------------------
// target 0 = RGBA32F (16 bytes per pixel; 12 when its constant alpha stays behind: RTX_OPT_GATHER_RGB, interleaved layout), 1 = RGBA8
inline size_t target_bytes(int target) { return target == 0 ? 16 : 4; }
inline size_t bytes_moved(int width, size_t rows, int target, bool rgb) { return rows * static_cast<size_t>(width) * (target == 0 && rgb ? size_t(12) : target_bytes(target)); }
// the buffer set (0 / 1) frame number `frame_no` uses: the gather of frame k overlaps the trace of frame k + 1, which uses the other set
inline int buffer_set(unsigned long long frame_no) { return static_cast<int>(frame_no & 1ull); }

// ---- first contact between ranks in separate processes (rtx_create_rank) -------------------------------------------------------------
// Every rank decides the byte counts of its paired ncclSend / ncclRecv on its own, from ITS copy of the frame configuration: frame size,
// rank count, band layout and rows, the contiguous split, which colour targets travel and whether the float target travels without its
// alpha. Two ranks that disagree (an option set on one process only) would hang or exchange garbage. So the configuration is digested to
// 16 bytes, and whenever a rank's digest differs from the one it last had confirmed, the ranks compare digests BEFORE any band travels:
// every peer sends its digest to rank 0, rank 0 answers each with a verdict (rtx_capi.cpp config_handshake; fixed-size messages, so
// they pair whatever the configurations are). FNV-1a over the fields, two seeds.
struct FrameConfig {
    int width = 0, height = 0, n_ranks = 0, band_rows = 0, band_layout = 0, gather_targets = 0, gather_rgb = 0, loopback = 0;
    std::vector<int> split_rows;      // contiguous layouts only (empty for the interleaved one)
};
struct ConfigDigest {
    unsigned long long a = 0, b = 0;
    bool operator==(const ConfigDigest& o) const { return a == o.a && b == o.b; }
    bool operator!=(const ConfigDigest& o) const { return !(*this == o); }
};
inline ConfigDigest config_digest(const FrameConfig& c)
{
    std::vector<int> v{c.width, c.height, c.n_ranks, c.band_rows, c.band_layout, c.gather_targets, c.gather_rgb, c.loopback, static_cast<int>(c.split_rows.size())};
    if (c.band_layout != 0) v.insert(v.end(), c.split_rows.begin(), c.split_rows.end());
    ConfigDigest d;
    d.a = 1469598103934665603ull; d.b = 0x9e3779b97f4a7c15ull;
    for (int x : v)
        for (int k = 0; k < 4; k++) {
            const unsigned long long byte = (static_cast<unsigned int>(x) >> (8 * k)) & 0xffu;
            d.a = (d.a ^ byte) * 1099511628211ull;
            d.b = (d.b ^ (byte + 0x51ull)) * 0x100000001b3ull + (d.b >> 29);
        }
    return d;
}
// rank 0's side of the handshake: digests[r] = what rank r sent (digests[0] = its own). Returns the first rank that disagrees, -1 if none.
inline int config_first_mismatch(const std::vector<ConfigDigest>& digests)
{
    for (size_t r = 1; r < digests.size(); r++)
        if (digests[r] != digests[0]) return static_cast<int>(r);
    return -1;
}
// A wait on the transfer stream that cannot hang the process: `done()` is polled (hipStreamQuery in the product, a counter in the host
// tests) until it reports completion or `timeout_ms` have passed on `now_ms()`; `nap()` yields between polls. timeout_ms <= 0: wait for ever
// (the behaviour of rounds 1-5). Returns true when the work completed.
template <class Done, class Now, class Nap>
inline bool bounded_wait(Done done, Now now_ms, Nap nap, double timeout_ms)
{
    const double t0 = now_ms();
    for (;;) {
        if (done()) return true;
        if (timeout_ms > 0.0 && now_ms() - t0 > timeout_ms) return false;
        nap();
    }
}

}  // namespace rtbands
