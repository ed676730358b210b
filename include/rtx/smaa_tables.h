// smaa_tables.h -- the two look-up tables of SMAA 1x, computed from their published construction (header-only, host code, no HIP).
//
// What this replaces: the reference uploads two precomputed byte arrays, `areaTexBytes` (160 x 560 texels of RG8, src/AreaTex.h:33-43,
// uploaded by SMAA_Builder::load_area_texture, src/SMAA_Builder.h:51-66) and `searchTexBytes` (64 x 16 texels of R8, src/SearchTex.h,
// SMAA_Builder.h:68-83). This repository stores neither array. Both are pure functions of a handful of published constants -- Jimenez,
// Echevarria, Sousa, Gutierrez, "SMAA: Enhanced Subpixel Morphological Antialiasing", Eurographics 2012, sections 3.2-3.5 and 4, and the
// table-building recipe the authors distribute with the shader -- so they are generated here on first use:
//
//   area table    for every crossing-edge pattern and every pair of distances (left, right) to the ends of a line of edges, the
//                 coverage of the pixel by the re-vectorised line on either side of the edge:
//                 * orthogonal lines: 16 patterns (4 crossing-edge bits), exact trapezoid / two-triangle areas under the straight
//                   segments that join the line's ends (offsets +-0.5 at an end with a crossing edge, 0 at mid-line for L shapes);
//                   U shapes shorter than 32 pixels are smoothed towards sqrt(2a)/2; texel i stands for distance i^2 (the shader
//                   looks it up with sqrt(d), SMAA.h:1212-1217), 16 x 16 texels per pattern, 5 x 5 pattern slots (slot = 4 e1 /
//                   4 e2 with e in {0, 0.25, 0.75, 1}: slot 2 stays empty);
//                 * diagonal lines: 16 patterns, coverage by brute-force sampling (30 x 30 points per pixel) of the half-plane of the
//                   re-vectorised diagonal, averaged over the two possible endings where the ending is unknown; 20 x 20 texels per
//                   pattern, 4 x 4 slots;
//                 * each repeated for the sub-sample offsets of the temporal / multisampled modes (7 orthogonal, 5 diagonal); SMAA 1x
//                   reads offset 0 only, but the table is the whole 160 x 560 array so that it is interchangeable with the reference's.
//                 Bytes are int(255 * area) (truncation).
//   search table  for a bilinear fetch of four edge texels at offset (-0.25, -0.125) -- one fetch that tells, for the end of a search,
//                 how many of the last two pixels still belong to the line (0, 1, 2 -> bytes 0, 127, 254): 66 x 33 entries cropped to
//                 64 x 16 and flipped vertically.
//
// Pinned: tests/test_smaa_tables.py compares the result with the reference's two arrays byte for byte wherever /root/reference exists
// (all 179 200 + 1 024 bytes equal), and with sha256 sums committed from that comparison everywhere else.
//
// Arithmetic: IEEE double throughout, no contraction (the package is built with -ffp-contract=off); every quantity below is a short
// sum / product / quotient of small dyadic numbers or one sqrt, so the results do not depend on the platform's libm beyond sqrt and modf.
#ifndef RTX_SMAA_TABLES_H_
#define RTX_SMAA_TABLES_H_

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace rtx_smaa {

enum { AREA_W = 160, AREA_H = 560, SEARCH_W = 64, SEARCH_H = 16, AREA_BYTES = AREA_W * AREA_H * 2, SEARCH_BYTES = SEARCH_W * SEARCH_H };

namespace detail {

struct V2 { double x, y; };
inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
inline V2 half(V2 a) { return {a.x / 2.0, a.y / 2.0}; }
inline double lerp(double a, double b, double p) { return a + (b - a) * p; }
inline double saturate(double v) { return v < 0.0 ? 0.0 : v > 1.0 ? 1.0 : v; }

// area between the x axis and the segment p1 -> p2 over the pixel column [x, x + 1], split into (below the axis, above the axis)
inline V2 column_area(V2 p1, V2 p2, double x)
{
    const V2 d = {p2.x - p1.x, p2.y - p1.y};
    const double x1 = x, x2 = x + 1.0;
    const double y1 = p1.y + d.y * (x1 - p1.x) / d.x;
    const double y2 = p1.y + d.y * (x2 - p1.x) / d.x;
    const bool inside = (x1 >= p1.x && x1 < p2.x) || (x2 > p1.x && x2 <= p2.x);
    if (!inside) return {0.0, 0.0};
    const bool trapezoid = std::copysign(1.0, y1) == std::copysign(1.0, y2) || std::fabs(y1) < 1e-4 || std::fabs(y2) < 1e-4;
    if (trapezoid) {
        const double a = (y1 + y2) / 2.0;
        return a < 0.0 ? V2{std::fabs(a), 0.0} : V2{0.0, std::fabs(a)};
    }
    // the segment crosses the axis inside the column: two triangles
    const double xc = -p1.y * d.x / d.y + p1.x;
    double ipart;
    const double frac = std::modf(xc, &ipart);
    const double a1 = xc > p1.x ? y1 * frac / 2.0 : 0.0;
    const double a2 = xc < p2.x ? y2 * (1.0 - frac) / 2.0 : 0.0;
    const double a = std::fabs(a1) > std::fabs(a2) ? a1 : -a2;
    return a < 0.0 ? V2{std::fabs(a1), std::fabs(a2)} : V2{std::fabs(a2), std::fabs(a1)};
}

// short U shapes: blend the exact areas towards sqrt(2a)/2 (paper section 3.3, "smoothing")
inline void smooth_u(double d, V2& a1, V2& a2)
{
    const V2 b1 = {std::sqrt(a1.x * 2.0) * 0.5, std::sqrt(a1.y * 2.0) * 0.5};
    const V2 b2 = {std::sqrt(a2.x * 2.0) * 0.5, std::sqrt(a2.y * 2.0) * 0.5};
    const double p = saturate(d / 32.0);
    a1 = {lerp(b1.x, a1.x, p), lerp(b1.y, a1.y, p)};
    a2 = {lerp(b2.x, a2.x, p), lerp(b2.y, a2.y, p)};
}

// orthogonal pattern: bit 0 = crossing edge below the left end, bit 1 = below the right end, bit 2 = above the left end,
// bit 3 = above the right end
inline V2 area_ortho(int pattern, int left, int right, double offset)
{
    const double d = left + right + 1;
    const double o1 = 0.5 + offset, o2 = 0.5 + offset - 1.0;
    const V2 L1 = {0.0, o1}, L2 = {0.0, o2}, M = {d / 2.0, 0.0}, R1 = {d, o1}, R2 = {d, o2};
    const double x = left;
    switch (pattern) {
        case 1: return left <= right ? column_area(L2, M, x) : V2{0.0, 0.0};   // L shapes are offset on the crossing-edge side only
        case 2: return left >= right ? column_area(M, R2, x) : V2{0.0, 0.0};
        case 3: { V2 a1 = column_area(L2, M, x), a2 = column_area(M, R2, x); smooth_u(d, a1, a2); return a1 + a2; }
        case 4: return left <= right ? column_area(L1, M, x) : V2{0.0, 0.0};
        case 6:   // Z shape; with a sub-sample offset, blended with the two partially offset L shapes it decays into at the search limit
            if (std::fabs(offset) > 0.0) return half(column_area(L1, R2, x) + (column_area(L1, M, x) + column_area(M, R2, x)));
            return column_area(L1, R2, x);
        case 7: return column_area(L1, R2, x);
        case 8: return left >= right ? column_area(M, R1, x) : V2{0.0, 0.0};
        case 9:
            if (std::fabs(offset) > 0.0) return half(column_area(L2, R1, x) + (column_area(L2, M, x) + column_area(M, R1, x)));
            return column_area(L2, R1, x);
        case 11: return column_area(L2, R1, x);
        case 12: { V2 a1 = column_area(L1, M, x), a2 = column_area(M, R1, x); smooth_u(d, a1, a2); return a1 + a2; }
        case 13: return column_area(L2, R1, x);
        case 14: return column_area(L1, R2, x);
        default: return {0.0, 0.0};   // 0: no crossing edge; 5, 10, 15: crossing edges on both sides of an end
    }
}

// fraction of the 30 x 30 sample points of the unit pixel at `p` that lie on the positive side of the line p1 -> p2
inline double sampled_coverage(V2 p1, V2 p2, V2 p)
{
    const int S = 30;
    if (p1.x == p2.x && p1.y == p2.y) return 1.0;
    const double xm = (p1.x + p2.x) / 2.0, ym = (p1.y + p2.y) / 2.0;
    const double a = p2.y - p1.y, b = p1.x - p2.x;
    int count = 0;
    for (int i = 0; i < S; i++)
        for (int j = 0; j < S; j++) {
            const double qx = p.x + i / static_cast<double>(S - 1), qy = p.y + j / static_cast<double>(S - 1);
            if (a * (qx - xm) + b * (qy - ym) > 0.0) count++;
        }
    return count / static_cast<double>(S * S);
}

// diagonal pattern e = (e1, e2), slot of the left / right end: an end with a crossing edge (slot > 0) is moved by the sub-sample offset
inline V2 diag_pair(int e1, int e2, V2 p1, V2 p2, int left, V2 offset)
{
    if (e1 > 0) p1 = p1 + offset;
    if (e2 > 0) p2 = p2 + offset;
    const double l = left;
    const double a1 = sampled_coverage(p1, p2, V2{1.0 + l, 0.0 + l});
    const double a2 = sampled_coverage(p1, p2, V2{1.0 + l, 1.0 + l});
    return {1.0 - a1, a2};
}

inline V2 area_diag(int pattern, int e1, int e2, int left, int right, V2 offset)
{
    // start corner (near the pixel) and end corner (relative to (d, d)) of the re-vectorised diagonal: one entry where both ends are
    // known, two -- averaged -- where a pattern leaves an ending open (paper section 3.4)
    struct Seg { double ax, ay, bx, by; };
    static const Seg kSeg[16][2] = {
        {{1, 1, 1, 1}, {1, 0, 1, 0}}, {{1, 0, 0, 0}, {1, 0, 1, 0}}, {{0, 0, 1, 0}, {1, 0, 1, 0}}, {{1, 0, 1, 0}, {-1, 0, 0, 0}},
        {{1, 1, 0, 0}, {1, 1, 1, 0}}, {{1, 1, 0, 0}, {1, 0, 1, 0}}, {{1, 1, 1, 0}, {-1, 0, 0, 0}}, {{1, 1, 1, 0}, {1, 0, 1, 0}},
        {{0, 0, 1, 1}, {1, 0, 1, 1}}, {{1, 0, 1, 1}, {-1, 0, 0, 0}}, {{0, 0, 1, 1}, {1, 0, 1, 0}}, {{1, 0, 1, 1}, {1, 0, 1, 0}},
        {{1, 1, 1, 1}, {-1, 0, 0, 0}}, {{1, 1, 1, 1}, {1, 0, 1, 1}}, {{1, 1, 1, 1}, {1, 1, 1, 0}}, {{1, 1, 1, 1}, {1, 0, 1, 0}}};
    const double d = left + right + 1;
    const Seg& s0 = kSeg[pattern][0];
    const Seg& s1 = kSeg[pattern][1];
    const V2 r0 = diag_pair(e1, e2, V2{s0.ax, s0.ay}, V2{s0.bx + d, s0.by + d}, left, offset);
    if (s1.ax < 0.0) return r0;
    const V2 r1 = diag_pair(e1, e2, V2{s1.ax, s1.ay}, V2{s1.bx + d, s1.by + d}, left, offset);
    return half(r0 + r1);
}

inline uint8_t to_byte(double v) { const int b = static_cast<int>(255.0 * v); return static_cast<uint8_t>(b < 0 ? 0 : b > 255 ? 255 : b); }

}  // namespace detail

// area: AREA_H rows of AREA_W texels of RG8, row 0 first -- the form of the reference's areaTexBytes (src/AreaTex.h:33-43)
inline void generate_area_table(uint8_t* out)
{
    using namespace detail;
    std::memset(out, 0, AREA_BYTES);
    static const double kOrthoOffsets[7] = {0.0, -0.25, 0.25, -0.125, 0.125, -0.375, 0.375};
    static const V2 kDiagOffsets[5] = {{0.0, 0.0}, {0.25, -0.25}, {-0.25, 0.25}, {0.125, -0.125}, {-0.125, 0.125}};
    // slot of a pattern in the 5 x 5 (orthogonal) / 4 x 4 (diagonal) grid: what the shader computes from the crossing-edge fetches
    // (round(4 e) with e in {0, 0.25, 0.75, 1}, SMAA.h:1218-1225; e1 + 2 e2 style sums for the diagonals, SMAA.h:957-1010)
    static const int kOrthoSlot[16][2] = {{0, 0}, {3, 0}, {0, 3}, {3, 3}, {1, 0}, {4, 0}, {1, 3}, {4, 3}, {0, 1}, {3, 1}, {0, 4}, {3, 4}, {1, 1}, {4, 1}, {1, 4}, {4, 4}};
    static const int kDiagSlot[16][2] = {{0, 0}, {1, 0}, {0, 2}, {1, 2}, {2, 0}, {3, 0}, {2, 2}, {3, 2}, {0, 1}, {1, 1}, {0, 3}, {1, 3}, {2, 1}, {3, 1}, {2, 3}, {3, 3}};
    auto put = [&](int x, int y, V2 a) {
        uint8_t* t = out + (static_cast<size_t>(y) * AREA_W + x) * 2;
        t[0] = to_byte(a.x);
        t[1] = to_byte(a.y);
    };
    for (int o = 0; o < 7; o++)
        for (int pat = 0; pat < 16; pat++)
            for (int y = 0; y < 16; y++)
                for (int x = 0; x < 16; x++)
                    put(kOrthoSlot[pat][0] * 16 + x, o * 80 + kOrthoSlot[pat][1] * 16 + y, area_ortho(pat, x * x, y * y, kOrthoOffsets[o]));
    for (int o = 0; o < 5; o++)
        for (int pat = 0; pat < 16; pat++)
            for (int y = 0; y < 20; y++)
                for (int x = 0; x < 20; x++)
                    put(80 + kDiagSlot[pat][0] * 20 + x, o * 80 + kDiagSlot[pat][1] * 20 + y,
                        area_diag(pat, kDiagSlot[pat][0], kDiagSlot[pat][1], x, y, kDiagOffsets[o]));
}

// search: SEARCH_H rows of SEARCH_W texels of R8, row 0 first -- the form of the reference's searchTexBytes (src/SearchTex.h)
inline void generate_search_table(uint8_t* out)
{
    // a bilinear fetch at (-0.25, -0.125) of four edge bits (e0 e1 / e2 e3) takes one of 16 distinct values k/32; invert that
    auto bilinear = [](const int e[4]) {
        auto lerp = [](double a, double b, double p) { return a + (b - a) * p; };
        return lerp(lerp(e[0], e[1], 1.0 - 0.25), lerp(e[2], e[3], 1.0 - 0.25), 1.0 - 0.125);
    };
    int decode[33][4];
    bool known[33] = {false};
    for (int m = 0; m < 16; m++) {
        const int e[4] = {(m >> 3) & 1, (m >> 2) & 1, (m >> 1) & 1, m & 1};
        const double v = bilinear(e) * 32.0;
        const int k = static_cast<int>(v);
        if (static_cast<double>(k) == v && k >= 0 && k <= 32) { known[k] = true; std::memcpy(decode[k], e, sizeof e); }
    }
    uint8_t img[33][66];
    std::memset(img, 0, sizeof img);
    for (int x = 0; x < 33; x++)
        for (int y = 0; y < 33; y++) {
            if (!known[x] || !known[y]) continue;
            const int* left = decode[x];
            const int* top = decode[y];
            int dl = 0, dr = 0;
            if (top[3] == 1) dl++;                                                  // there is an edge: the line goes on
            if (dl == 1 && top[2] == 1 && left[1] != 1 && left[3] != 1) dl++;        // another one and no crossing edge: one more
            if (top[3] == 1 && left[1] != 1 && left[3] != 1) dr++;
            if (dr == 1 && top[2] == 1 && left[0] != 1 && left[2] != 1) dr++;
            img[y][x] = static_cast<uint8_t>(127 * dl);
            img[y][33 + x] = static_cast<uint8_t>(127 * dr);
        }
    for (int r = 0; r < SEARCH_H; r++)   // rows 17..32 of the 66 x 33 image, flipped vertically; columns 0..63
        std::memcpy(out + static_cast<size_t>(r) * SEARCH_W, &img[32 - r][0], SEARCH_W);
}

}  // namespace rtx_smaa

#endif  // RTX_SMAA_TABLES_H_
