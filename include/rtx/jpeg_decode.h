// jpeg_decode.h -- a JPEG reader for the C++ shim (asset ingestion).
//
// The reference decodes its texture files with stb_image v2.25 (GLWrapper.cpp:293,325: stbi_load(path,&w,&h,&c,0)); three
// of its five default textures and all six sky-box faces are baseline JPEGs (4:4:4 and 4:2:0). Unlike PNG, the JPEG
// standard (ITU-T T.81) does not fix the decoder's arithmetic, so "the same texels as the reference" means stb_image's
// choices for the three lossy-side steps, which this header restates in its own code:
//   * inverse DCT: the Loeffler-Ligtenberg-Moschytz factorisation in 12-bit fixed point (the jidctint scheme): column
//     pass rounded to 2 fractional bits ((x + 512) >> 10), row pass rounded to integers with the +128 level shift folded
//     in ((x + 65536 + (128 << 17)) >> 17), clamped to 0..255;
//   * chroma up-sampling: 2x horizontally and/or vertically with the 3:1 triangle filter (rounding +2 >> 2 for one axis,
//     +8 >> 4 for both, edge samples replicated -- including stb_image's asymmetric right edge in the horizontal-only
//     case); other ratios by sample replication; the row nearer to the output row is the "3" row;
//   * YCbCr -> RGB in 20-bit fixed point with the constants 1.40200, 0.71414, 0.34414, 1.77200 quantised to 12 bits, the
//     Cb term of green truncated to its upper 16 bits, rounding +0.5, arithmetic shift, clamp.
// Entropy decoding, marker syntax, progressive scan semantics follow T.81 (sections B, F.2, G.1.2, G.2).
// Output convention = stb_image's with req_comp = 0: 8 bits per channel, interleaved, top row first; 1 channel for
// single-component files, 3 for three- and four-component files (CMYK / YCCK with the Adobe marker are folded into RGB
// like stb_image does); RGB-coded files (component ids 'R','G','B', or an Adobe transform 0 without JFIF) are copied.
// Supported: SOF0/SOF1 (sequential Huffman, 8-bit) and SOF2 (progressive Huffman), interleaved and non-interleaved
// scans, restart intervals, 8- and 16-bit quantisation tables. Not supported (nullptr): arithmetic coding, lossless,
// hierarchical, 12-bit samples.
// Checked bit for bit against the reference's stb_image on generated files and on the reference's nine JPEG assets
// (tests/test_jpeg_decode.py; fixtures under tests/golden/jpeg/).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace rtx_jpeg {

// zig-zag position -> natural (row-major) index; the tail guards k running past 63 on damaged streams
static const uint8_t kNatural[64 + 16] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct HuffTable {           // T.81 annex C (code generation) and F.2.2.3 (decoding)
    bool defined = false;
    uint8_t sym[256];
    uint16_t look[256];      // first 8 bits of the stream -> (length << 8) | symbol, 0 if the code is longer than 8 bits
    int32_t maxcode[18];     // largest code of each length, left-aligned to 16 bits, +1 (exclusive bound)
    int32_t first_index[17]; // index into sym of the first code of each length minus that first code
    bool build(const uint8_t* counts /* 16 */, const uint8_t* symbols, int n_symbols)
    {
        std::memcpy(sym, symbols, static_cast<size_t>(n_symbols));
        std::memset(look, 0, sizeof look);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            first_index[len] = k - code;
            for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
                if (len <= 8) {
                    const int lo = code << (8 - len), hi = lo + (1 << (8 - len));
                    for (int v = lo; v < hi && v < 256; ++v) look[v] = static_cast<uint16_t>((len << 8) | sym[k]);
                }
            }
            if (code > (1 << len)) return false;   // over-subscribed
            maxcode[len] = code << (16 - len);
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        defined = true;
        return true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;          // tables selected by the current scan
    int x = 0, y = 0;            // samples of this component that are part of the image
    int w2 = 0, h2 = 0;          // plane size, padded to whole MCUs
    int pred = 0;                // DC predictor
    std::vector<uint8_t> plane;  // reconstructed samples (w2 x h2)
    std::vector<int16_t> coeff;  // progressive: quantised coefficients, 64 per block, (w2/8) blocks per row
};

class Decoder {
public:
    Decoder(const uint8_t* d, size_t n) : begin(d), p(d), end(d + n) {}

    unsigned char* run(int* w_out, int* h_out, int* ch_out)
    {
        if (!(byte() == 0xFF && byte() == 0xD8)) return nullptr;
        for (;;) {
            const int m = next_marker();
            if (m < 0) {
                if (!have_frame && p < end) continue;               // stray bytes between the segments of the header
                return nullptr;
            }
            if (m == 0xD9) break;                                   // EOI
            if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
                if (have_frame || !frame_header(m == 0xC2)) return nullptr;
            } else if (m == 0xDA) {
                if (!have_frame || !scan_header() || !scan_data()) return nullptr;
                if (pending_marker < 0) {                           // entropy data ended without running into a marker
                    while (p < end && *p != 0xFF) ++p;
                }
            } else if (m == 0xDC) {                                 // DNL: must repeat the frame's height
                const int len = be16();
                const int nl = be16();
                if (len != 4 || nl != height) return nullptr;
            } else if (!table_or_skip(m)) {
                return nullptr;
            }
        }
        if (!have_frame) return nullptr;
        if (progressive) reconstruct_progressive();
        return to_pixels(w_out, h_out, ch_out);
    }

private:
    // ---- byte level ---------------------------------------------------------------------------------------------
    const uint8_t* begin;
    const uint8_t* p;
    const uint8_t* end;
    int byte() { return p < end ? *p++ : 0; }
    int be16() { const int a = byte(); return (a << 8) | byte(); }
    // the marker code that follows (fill bytes skipped); -1 when the next byte is not 0xFF
    int next_marker()
    {
        if (pending_marker >= 0) { const int m = pending_marker; pending_marker = -1; return m; }
        int c = byte();
        if (c != 0xFF) return -1;
        while (c == 0xFF) c = byte();
        return c;
    }

    // ---- frame state --------------------------------------------------------------------------------------------
    bool have_frame = false, progressive = false, jfif = false;
    int adobe_transform = -1, rgb_ids = 0;
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
    int restart_interval = 0;
    Component comp[4];
    uint16_t quant[4][64];   // natural order
    HuffTable dc_tab[4], ac_tab[4];

    bool table_or_skip(int m)
    {
        if (m == 0xDD) {                       // DRI
            if (be16() != 4) return false;
            restart_interval = be16();
            return true;
        }
        if (m == 0xDB) {                       // DQT
            int len = be16() - 2;
            while (len > 0) {
                const int pq = byte();
                const int wide = pq >> 4, t = pq & 15;
                if (wide > 1 || t > 3) return false;
                for (int i = 0; i < 64; ++i) quant[t][kNatural[i]] = static_cast<uint16_t>(wide ? be16() : byte());
                len -= wide ? 129 : 65;
            }
            return len == 0;
        }
        if (m == 0xC4) {                       // DHT
            int len = be16() - 2;
            while (len > 0) {
                const int tc_th = byte();
                const int cls = tc_th >> 4, t = tc_th & 15;
                if (cls > 1 || t > 3) return false;
                uint8_t counts[16], symbols[256];
                int n = 0;
                for (int i = 0; i < 16; ++i) { counts[i] = static_cast<uint8_t>(byte()); n += counts[i]; }
                if (n > 256) return false;
                for (int i = 0; i < n; ++i) symbols[i] = static_cast<uint8_t>(byte());
                if (!(cls ? ac_tab : dc_tab)[t].build(counts, symbols, n)) return false;
                len -= 17 + n;
            }
            return len == 0;
        }
        if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {   // APPn, COM
            int len = be16();
            if (len < 2) return false;
            len -= 2;
            if (m == 0xE0 && len >= 5 && end - p >= 5 && std::memcmp(p, "JFIF\0", 5) == 0) jfif = true;
            if (m == 0xEE && len >= 12 && end - p >= 12 && std::memcmp(p, "Adobe\0", 6) == 0) adobe_transform = p[11];
            if (end - p < len) return false;
            p += len;
            return true;
        }
        return false;                           // arithmetic / lossless / hierarchical frames and anything unknown
    }

    bool frame_header(bool prog)
    {
        const int len = be16();
        if (byte() != 8) return false;          // sample precision
        height = be16();
        width = be16();
        ncomp = byte();
        if (height <= 0 || width <= 0 || !(ncomp == 1 || ncomp == 3 || ncomp == 4) || len != 8 + 3 * ncomp) return false;
        if (static_cast<uint64_t>(width) * static_cast<uint64_t>(height) > (1ull << 30)) return false;
        // a block costs at least one bit of entropy-coded data per scan: a header that promises more than 512 pixels per
        // file byte is damaged, and must not size the allocations below
        if (static_cast<uint64_t>(width) * static_cast<uint64_t>(height) > 1024ull * static_cast<uint64_t>(end - begin) + 65536ull) return false;
        static const char rgb[3] = {'R', 'G', 'B'};
        for (int i = 0; i < ncomp; ++i) {
            Component& c = comp[i];
            c.id = byte();
            if (ncomp == 3 && c.id == rgb[i]) ++rgb_ids;
            const int hv = byte();
            c.h = hv >> 4;
            c.v = hv & 15;
            c.tq = byte();
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return false;
            if (c.h > hmax) hmax = c.h;
            if (c.v > vmax) vmax = c.v;
        }
        // every component's sampling factors must divide the maxima: the up-sampler works with the integer ratios hmax/h,
        // vmax/v, and e.g. H = (3,2,1) would make it read `width` samples from plane rows that are narrower than that.
        // (stb_image v2.25, the reference's decoder, accepts such frames and does read across the row ends; there is no
        // defined output to reproduce, so this reader refuses them, as later stb_image releases do.)
        for (int i = 0; i < ncomp; ++i)
            if (hmax % comp[i].h != 0 || vmax % comp[i].v != 0) return false;
        mcus_x = (width + 8 * hmax - 1) / (8 * hmax);
        mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < ncomp; ++i) {
            Component& c = comp[i];
            c.x = (width * c.h + hmax - 1) / hmax;
            c.y = (height * c.v + vmax - 1) / vmax;
            c.w2 = mcus_x * c.h * 8;
            c.h2 = mcus_y * c.v * 8;
            c.plane.assign(static_cast<size_t>(c.w2) * static_cast<size_t>(c.h2), 0);
            if (prog) c.coeff.assign(static_cast<size_t>(c.w2) * static_cast<size_t>(c.h2), 0);
        }
        progressive = prog;
        have_frame = true;
        return true;
    }

    // ---- scan state ---------------------------------------------------------------------------------------------
    int scan_n = 0, scan_comp[4] = {0, 0, 0, 0};
    int ss = 0, se = 63, ah = 0, al = 0, eob_run = 0;
    uint32_t acc = 0;        // bit accumulator, next bit = MSB
    int nbits = 0;
    int pending_marker = -1; // a marker the bit reader ran into (the entropy-coded segment is over: zeros follow)

    bool scan_header()
    {
        const int len = be16();
        scan_n = byte();
        if (scan_n < 1 || scan_n > 4 || scan_n > ncomp || len != 6 + 2 * scan_n) return false;
        for (int k = 0; k < scan_n; ++k) {
            const int id = byte(), tables = byte();
            int which = -1;
            for (int i = 0; i < ncomp; ++i) if (comp[i].id == id) { which = i; break; }
            if (which < 0) return false;
            comp[which].td = tables >> 4;
            comp[which].ta = tables & 15;
            if (comp[which].td > 3 || comp[which].ta > 3) return false;
            scan_comp[k] = which;
        }
        ss = byte();
        se = byte();
        const int a = byte();
        ah = a >> 4;
        al = a & 15;
        if (progressive) {
            if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) return false;
            if (ss == 0 && se != 0) return false;                  // a scan carries either the DC or an AC band
        } else {
            if (ss != 0 || ah != 0 || al != 0) return false;
            se = 63;
        }
        return true;
    }

    void fill()
    {
        while (nbits <= 24) {
            int b = 0;
            if (pending_marker < 0 && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    int c = byte();
                    while (c == 0xFF) c = byte();       // fill bytes
                    if (c != 0) { pending_marker = c; b = 0; }
                }
            }
            acc |= static_cast<uint32_t>(b) << (24 - nbits);
            nbits += 8;
        }
    }
    int bits(int n)          // n in 0..16
    {
        if (n == 0) return 0;
        if (nbits < n) fill();
        const int v = static_cast<int>(acc >> (32 - n));
        acc <<= n;
        nbits -= n;
        return v;
    }
    int signed_bits(int n)   // T.81 F.2.2.1 EXTEND(RECEIVE(n), n)
    {
        const int v = bits(n);
        return (n && v < (1 << (n - 1))) ? v - (1 << n) + 1 : v;
    }
    int symbol(const HuffTable& t)   // -1: no such code
    {
        if (nbits < 16) fill();
        const int e = t.look[acc >> 24];
        if (e) {
            const int len = e >> 8;
            acc <<= len;
            nbits -= len;
            return e & 255;
        }
        const int32_t top = static_cast<int32_t>(acc >> 16);
        int len = 9;
        while (top >= t.maxcode[len]) ++len;
        if (len > 16) return -1;
        const int idx = t.first_index[len] + (top >> (16 - len));
        if (idx < 0 || idx > 255) return -1;
        acc <<= len;
        nbits -= len;
        return t.sym[idx];
    }
    void restart_state()
    {
        acc = 0;
        nbits = 0;
        pending_marker = -1;
        eob_run = 0;
        for (int i = 0; i < 4; ++i) comp[i].pred = 0;
    }
    // after each restart interval: the segment must end in RSTn; false = stop this scan here
    bool at_restart()
    {
        if (nbits < 24) fill();
        if (pending_marker < 0xD0 || pending_marker > 0xD7) return false;
        restart_state();
        return true;
    }

    // ---- block decoders -----------------------------------------------------------------------------------------
    static inline int wrap_add(int a, int b) { return static_cast<int32_t>(static_cast<uint32_t>(a) + static_cast<uint32_t>(b)); }
    static inline int wrap_mul(int a, int b) { return static_cast<int32_t>(static_cast<uint32_t>(a) * static_cast<uint32_t>(b)); }
    bool block_sequential(Component& c, int16_t* blk)
    {
        const HuffTable& hd = dc_tab[c.td];
        const HuffTable& ha = ac_tab[c.ta];
        const uint16_t* q = quant[c.tq];
        std::memset(blk, 0, 64 * sizeof(int16_t));
        const int t = symbol(hd);
        if (t < 0 || t > 15) return false;
        c.pred = wrap_add(c.pred, signed_bits(t));
        blk[0] = static_cast<int16_t>(wrap_mul(c.pred, q[0]));
        for (int k = 1; k < 64;) {
            const int rs = symbol(ha);
            if (rs < 0) return false;
            const int run = rs >> 4, size = rs & 15;
            if (size == 0) {
                if (rs != 0xF0) break;          // end of block
                k += 16;
            } else {
                k += run;
                const int nat = kNatural[k++];
                blk[nat] = static_cast<int16_t>(signed_bits(size) * q[nat]);
            }
        }
        return true;
    }
    bool block_dc_progressive(Component& c, int16_t* blk)
    {
        if (ah == 0) {                           // first pass: the DC value, scaled by the point transform
            const int t = symbol(dc_tab[c.td]);
            if (t < 0 || t > 15) return false;
            c.pred = wrap_add(c.pred, signed_bits(t));
            blk[0] = static_cast<int16_t>(wrap_mul(c.pred, 1 << al));
        } else if (bits(1)) {                    // refinement: one more bit
            blk[0] = static_cast<int16_t>(blk[0] + (1 << al));
        }
        return true;
    }
    void refine(int16_t& coef, int bit)          // correction bit for an already non-zero coefficient (G.1.2.3)
    {
        if (bits(1) && (coef & bit) == 0) coef = static_cast<int16_t>(coef > 0 ? coef + bit : coef - bit);
    }
    bool block_ac_progressive(Component& c, int16_t* blk)
    {
        const HuffTable& ha = ac_tab[c.ta];
        if (ah == 0) {                           // first pass over the band ss..se
            if (eob_run) { --eob_run; return true; }
            for (int k = ss; k <= se;) {
                const int rs = symbol(ha);
                if (rs < 0) return false;
                const int run = rs >> 4, size = rs & 15;
                if (size == 0) {
                    if (run < 15) {              // EOBn: this block and the next eob_run blocks have no more coefficients
                        eob_run = (1 << run) - 1;
                        if (run) eob_run += bits(run);
                        break;
                    }
                    k += 16;
                } else {
                    k += run;
                    blk[kNatural[k++]] = static_cast<int16_t>(signed_bits(size) * (1 << al));
                }
            }
            return true;
        }
        const int bit = 1 << al;                 // refinement pass
        if (eob_run) {
            --eob_run;
            for (int k = ss; k <= se; ++k) {
                int16_t& coef = blk[kNatural[k]];
                if (coef != 0) refine(coef, bit);
            }
            return true;
        }
        int k = ss;
        do {
            const int rs = symbol(ha);
            if (rs < 0) return false;
            int run = rs >> 4, value = 0;
            const int size = rs & 15;
            if (size == 0) {
                if (run < 15) {
                    eob_run = (1 << run) - 1;
                    if (run) eob_run += bits(run);
                    run = 64;                    // no new coefficient: only refine what is left of the band
                }
            } else {
                if (size != 1) return false;
                value = bits(1) ? bit : -bit;
            }
            while (k <= se) {
                int16_t& coef = blk[kNatural[k++]];
                if (coef != 0) {
                    refine(coef, bit);
                } else {
                    if (run == 0) { coef = static_cast<int16_t>(value); break; }
                    --run;
                }
            }
        } while (k <= se);
        return true;
    }

    // ---- inverse DCT --------------------------------------------------------------------------------------------
    // All sums and products wrap modulo 2^32 (damaged streams can carry any coefficient); the two's-complement reading of
    // the result is what the reference decoder's int arithmetic gives on every platform it runs on.
    typedef uint32_t U;
    static constexpr U fx(float v) { return static_cast<U>(static_cast<int>(v * 4096 + 0.5)); }
    static inline int sar(U v, int n) { return static_cast<int32_t>(v) >> n; }
    // one 8-point pass; in: 8 values, out: even-part sums a[0..3] (scaled by 4096) and odd-part sums o[0..3]; the caller adds
    // its rounding constant to a[] and forms a[i] +- o[3 - i]
    static inline void pass(const int in[8], U a[4], U o[4])
    {
        U s[8];
        for (int i = 0; i < 8; ++i) s[i] = static_cast<U>(in[i]);
        const U z = (s[2] + s[6]) * fx(0.5411961f);
        const U e2 = z + s[6] * fx(-1.847759065f);
        const U e3 = z + s[2] * fx(0.765366865f);
        const U e0 = (s[0] + s[4]) * 4096u, e1 = (s[0] - s[4]) * 4096u;
        a[0] = e0 + e3;
        a[3] = e0 - e3;
        a[1] = e1 + e2;
        a[2] = e1 - e2;
        const U z5 = (s[7] + s[3] + s[5] + s[1]) * fx(1.175875602f);
        const U z1 = z5 + (s[7] + s[1]) * fx(-0.899976223f);
        const U z2 = z5 + (s[5] + s[3]) * fx(-2.562915447f);
        const U z3 = (s[7] + s[3]) * fx(-1.961570560f);
        const U z4 = (s[5] + s[1]) * fx(-0.390180644f);
        o[0] = s[7] * fx(0.298631336f) + z1 + z3;
        o[1] = s[5] * fx(2.053119869f) + z2 + z4;
        o[2] = s[3] * fx(3.072711026f) + z2 + z3;
        o[3] = s[1] * fx(1.501321110f) + z1 + z4;
    }
    static inline uint8_t clamp8(int v) { return static_cast<uint8_t>(v < 0 ? 0 : (v > 255 ? 255 : v)); }
    static void idct(uint8_t* out, int stride, const int16_t* blk)
    {
        int mid[64];
        int s[8];
        U a[4], o[4];
        for (int col = 0; col < 8; ++col) {
            for (int i = 0; i < 8; ++i) s[i] = blk[8 * i + col];
            pass(s, a, o);
            for (int i = 0; i < 4; ++i) {
                const U e = a[i] + 512u;
                mid[8 * i + col] = sar(e + o[3 - i], 10);
                mid[8 * (7 - i) + col] = sar(e - o[3 - i], 10);
            }
        }
        for (int row = 0; row < 8; ++row) {
            pass(mid + 8 * row, a, o);
            uint8_t* d = out + static_cast<size_t>(row) * static_cast<size_t>(stride);
            for (int i = 0; i < 4; ++i) {
                const U e = a[i] + 65536u + (128u << 17);
                d[i] = clamp8(sar(e + o[3 - i], 17));
                d[7 - i] = clamp8(sar(e - o[3 - i], 17));
            }
        }
    }

    // ---- entropy-coded segment ----------------------------------------------------------------------------------
    bool scan_data()
    {
        for (int k = 0; k < scan_n; ++k) {
            const Component& c = comp[scan_comp[k]];
            const bool need_dc = !progressive || ss == 0, need_ac = !progressive || se > 0;
            if (need_dc && !(progressive && ah) && !dc_tab[c.td].defined) return false;
            if (need_ac && !ac_tab[c.ta].defined) return false;
        }
        if (progressive && ss > 0 && scan_n != 1) return false;   // AC bands are never interleaved
        restart_state();
        int todo = restart_interval ? restart_interval : 0x7fffffff;
        int16_t blk[64];
        if (scan_n == 1) {                       // non-interleaved: the component's own blocks, row by row
            Component& c = comp[scan_comp[0]];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3, per_row = c.w2 >> 3;
            for (int j = 0; j < bh; ++j) {
                for (int i = 0; i < bw; ++i) {
                    if (progressive) {
                        int16_t* q = c.coeff.data() + 64 * (static_cast<size_t>(j) * static_cast<size_t>(per_row) + static_cast<size_t>(i));
                        if (!(ss == 0 ? block_dc_progressive(c, q) : block_ac_progressive(c, q))) return false;
                    } else {
                        if (!block_sequential(c, blk)) return false;
                        idct(c.plane.data() + static_cast<size_t>(j) * 8 * static_cast<size_t>(c.w2) + static_cast<size_t>(i) * 8, c.w2, blk);
                    }
                    if (--todo <= 0) {
                        if (!at_restart()) return true;
                        todo = restart_interval;
                    }
                }
            }
            return true;
        }
        for (int j = 0; j < mcus_y; ++j) {       // interleaved: MCU by MCU, h x v blocks of each component
            for (int i = 0; i < mcus_x; ++i) {
                for (int k = 0; k < scan_n; ++k) {
                    Component& c = comp[scan_comp[k]];
                    const int per_row = c.w2 >> 3;
                    for (int by = 0; by < c.v; ++by) {
                        for (int bx = 0; bx < c.h; ++bx) {
                            const size_t col = static_cast<size_t>(i) * static_cast<size_t>(c.h) + static_cast<size_t>(bx);
                            const size_t row = static_cast<size_t>(j) * static_cast<size_t>(c.v) + static_cast<size_t>(by);
                            if (progressive) {
                                if (!block_dc_progressive(c, c.coeff.data() + 64 * (row * static_cast<size_t>(per_row) + col))) return false;
                            } else {
                                if (!block_sequential(c, blk)) return false;
                                idct(c.plane.data() + row * 8 * static_cast<size_t>(c.w2) + col * 8, c.w2, blk);
                            }
                        }
                    }
                }
                if (--todo <= 0) {
                    if (!at_restart()) return true;
                    todo = restart_interval;
                }
            }
        }
        return true;
    }

    void reconstruct_progressive()
    {
        int16_t blk[64];
        for (int n = 0; n < ncomp; ++n) {
            Component& c = comp[n];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3, per_row = c.w2 >> 3;
            const uint16_t* q = quant[c.tq];
            for (int j = 0; j < bh; ++j)
                for (int i = 0; i < bw; ++i) {
                    const int16_t* src = c.coeff.data() + 64 * (static_cast<size_t>(j) * static_cast<size_t>(per_row) + static_cast<size_t>(i));
                    for (int k = 0; k < 64; ++k) blk[k] = static_cast<int16_t>(src[k] * q[k]);
                    idct(c.plane.data() + static_cast<size_t>(j) * 8 * static_cast<size_t>(c.w2) + static_cast<size_t>(i) * 8, c.w2, blk);
                }
        }
    }

    // ---- up-sampling and colour ---------------------------------------------------------------------------------
    // one output row of a component: `near` is the plane row closer to the output row, `far` the other vertical neighbour
    static const uint8_t* upsample(uint8_t* out, const uint8_t* near, const uint8_t* far, int w, int hs, int vs)
    {
        if (hs == 1 && vs == 1) return near;
        if (hs == 1 && vs == 2) {
            for (int i = 0; i < w; ++i) out[i] = static_cast<uint8_t>((3 * near[i] + far[i] + 2) >> 2);
        } else if (hs == 2 && vs == 1) {
            if (w == 1) {
                out[0] = out[1] = near[0];
            } else {
                out[0] = near[0];
                out[1] = static_cast<uint8_t>((near[0] * 3 + near[1] + 2) >> 2);
                for (int i = 1; i < w - 1; ++i) {
                    const int n = 3 * near[i] + 2;
                    out[2 * i] = static_cast<uint8_t>((n + near[i - 1]) >> 2);
                    out[2 * i + 1] = static_cast<uint8_t>((n + near[i + 1]) >> 2);
                }
                out[2 * w - 2] = static_cast<uint8_t>((near[w - 2] * 3 + near[w - 1] + 2) >> 2);   // sic: stb_image's right edge
                out[2 * w - 1] = near[w - 1];
            }
        } else if (hs == 2 && vs == 2) {
            int cur = 3 * near[0] + far[0];
            out[0] = static_cast<uint8_t>((cur + 2) >> 2);
            if (w == 1) {
                out[1] = out[0];
            } else {
                for (int i = 1; i < w; ++i) {
                    const int prev = cur;
                    cur = 3 * near[i] + far[i];
                    out[2 * i - 1] = static_cast<uint8_t>((3 * prev + cur + 8) >> 4);
                    out[2 * i] = static_cast<uint8_t>((3 * cur + prev + 8) >> 4);
                }
                out[2 * w - 1] = static_cast<uint8_t>((cur + 2) >> 2);
            }
        } else {
            for (int i = 0; i < w; ++i)
                for (int j = 0; j < hs; ++j) out[i * hs + j] = near[i];
        }
        return out;
    }
    static constexpr int fx20(float v) { return static_cast<int>(v * 4096.0f + 0.5f) << 8; }
    static inline void ycc(uint8_t* rgb, int y, int cb, int cr)
    {
        const int base = (y << 20) + (1 << 19);
        cb -= 128;
        cr -= 128;
        const int r = base + cr * fx20(1.40200f);
        const int g = base + cr * -fx20(0.71414f) + static_cast<int>(static_cast<uint32_t>(cb * -fx20(0.34414f)) & 0xffff0000u);
        const int b = base + cb * fx20(1.77200f);
        rgb[0] = clamp8(r >> 20);
        rgb[1] = clamp8(g >> 20);
        rgb[2] = clamp8(b >> 20);
    }
    static inline uint8_t mul255(int a, int b) { const int t = a * b + 128; return static_cast<uint8_t>((t + (t >> 8)) >> 8); }

    unsigned char* to_pixels(int* w_out, int* h_out, int* ch_out)
    {
        const int out_ch = ncomp >= 3 ? 3 : 1;
        const bool plain_rgb = ncomp == 3 && (rgb_ids == 3 || (adobe_transform == 0 && !jfif));
        unsigned char* out = static_cast<unsigned char*>(std::malloc(static_cast<size_t>(width) * static_cast<size_t>(height) * static_cast<size_t>(out_ch)));
        if (!out) return nullptr;
        struct Row { int hs, vs, w_lores, ystep, ypos; const uint8_t* line0; const uint8_t* line1; std::vector<uint8_t> buf; };
        Row rows[4];
        for (int n = 0; n < ncomp; ++n) {
            Row& r = rows[n];
            r.hs = hmax / comp[n].h;
            r.vs = vmax / comp[n].v;
            r.w_lores = (width + r.hs - 1) / r.hs;
            r.ystep = r.vs >> 1;
            r.ypos = 0;
            r.line0 = r.line1 = comp[n].plane.data();
            r.buf.assign(static_cast<size_t>(r.w_lores) * static_cast<size_t>(r.hs) + 8, 0);
        }
        for (int j = 0; j < height; ++j) {
            const uint8_t* line[4] = {nullptr, nullptr, nullptr, nullptr};
            for (int n = 0; n < ncomp; ++n) {
                Row& r = rows[n];
                const bool lower = r.ystep >= (r.vs >> 1);   // the output row lies in the lower half of the source row pair
                line[n] = upsample(r.buf.data(), lower ? r.line1 : r.line0, lower ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
                if (++r.ystep >= r.vs) {
                    r.ystep = 0;
                    r.line0 = r.line1;
                    if (++r.ypos < comp[n].y) r.line1 += comp[n].w2;
                }
            }
            unsigned char* d = out + static_cast<size_t>(j) * static_cast<size_t>(width) * static_cast<size_t>(out_ch);
            if (ncomp == 1) {
                std::memcpy(d, line[0], static_cast<size_t>(width));
            } else if (ncomp == 3) {
                if (plain_rgb) {
                    for (int i = 0; i < width; ++i, d += 3) { d[0] = line[0][i]; d[1] = line[1][i]; d[2] = line[2][i]; }
                } else {
                    for (int i = 0; i < width; ++i, d += 3) ycc(d, line[0][i], line[1][i], line[2][i]);
                }
            } else if (adobe_transform == 0) {               // CMYK
                for (int i = 0; i < width; ++i, d += 3) {
                    const int k = line[3][i];
                    d[0] = mul255(line[0][i], k);
                    d[1] = mul255(line[1][i], k);
                    d[2] = mul255(line[2][i], k);
                }
            } else if (adobe_transform == 2) {               // YCCK
                for (int i = 0; i < width; ++i, d += 3) {
                    const int k = line[3][i];
                    ycc(d, line[0][i], line[1][i], line[2][i]);
                    d[0] = mul255(255 - d[0], k);
                    d[1] = mul255(255 - d[1], k);
                    d[2] = mul255(255 - d[2], k);
                }
            } else {
                for (int i = 0; i < width; ++i, d += 3) ycc(d, line[0][i], line[1][i], line[2][i]);
            }
        }
        *w_out = width;
        *h_out = height;
        *ch_out = out_ch;
        return out;
    }
};

// bytes (any file contents) -> malloc'ed interleaved 8-bit texels; nullptr on failure
inline unsigned char* decode_memory(const uint8_t* d, size_t n, int* w, int* h, int* channels)
{
    Decoder dec(d, n);
    return dec.run(w, h, channels);
}

inline unsigned char* decode_file(const char* path, int* w, int* h, int* channels)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) return nullptr;
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    size_t got;
    while ((got = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
    std::fclose(f);
    return decode_memory(buf.data(), buf.size(), w, h, channels);
}

}  // namespace rtx_jpeg
