// png_decode.h -- a small PNG reader for the C++ shim (asset ingestion).
//
// The reference decodes its texture files with stb_image (GLWrapper.cpp:293,325: stbi_load(path,&w,&h,&c,0)); two of
// its five default textures are PNGs (8-bit RGBA). This header reads PNG files without any third-party code so that a
// scene program built against the shim can load them as they are. Written from the specifications (PNG: ISO/IEC 15948,
// DEFLATE: RFC 1951, zlib: RFC 1950). Output convention = stb_image's with req_comp = 0:
//   * 8 bits per channel, interleaved, row 0 = top row of the file;
//   * channels: grey 1, grey+alpha 2, RGB 3, RGBA 4; palette images come out as RGB, or RGBA when a tRNS chunk exists;
//     a tRNS colour key on grey / RGB images adds an alpha channel (0 for the key colour, 255 elsewhere);
//   * 16-bit samples are reduced to their high byte; 1/2/4-bit grey samples are scaled to 0..255 (x255, x85, x17).
// Supported: all five colour types, bit depths 1, 2, 4, 8, 16, non-interlaced and Adam7-interlaced files.
// Checksums (chunk CRCs, Adler-32) are not verified, like stb_image.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace rtx_png {

// ---- DEFLATE (RFC 1951) --------------------------------------------------------------------------------------
struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int n = 0;
    bool bad = false;
    uint32_t bits(int k)   // k <= 16, LSB first
    {
        while (n < k) {
            if (p >= end) { bad = true; return 0; }
            acc |= static_cast<uint32_t>(*p++) << n;
            n += 8;
        }
        const uint32_t v = acc & ((1u << k) - 1u);
        acc >>= k;
        n -= k;
        return v;
    }
    void align_byte() { acc = 0; n = 0; }
};

struct Huffman {   // canonical code, decoded bit by bit over (count, symbol) tables -- section 3.2.2
    uint16_t count[16];
    uint16_t symbol[288];
    bool build(const uint8_t* lengths, int nsym)
    {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < nsym; i++) count[lengths[i]]++;
        count[0] = 0;
        int left = 1;
        for (int len = 1; len < 16; len++) {
            left = (left << 1) - count[len];
            if (left < 0) return false;   // over-subscribed
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int len = 1; len < 15; len++) offs[len + 1] = static_cast<uint16_t>(offs[len] + count[len]);
        for (int i = 0; i < nsym; i++)
            if (lengths[i]) symbol[offs[lengths[i]]++] = static_cast<uint16_t>(i);
        return true;
    }
    int decode(BitReader& br) const
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; len++) {
            code |= static_cast<int>(br.bits(1));
            if (br.bad) return -1;
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

inline bool inflate(const uint8_t* src, size_t n, std::vector<uint8_t>& out)
{
    static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const uint8_t cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    if (n < 2) return false;
    BitReader br{src + 2, src + n};   // skip the zlib header (CMF, FLG)
    if ((src[0] & 15) != 8 || (src[1] & 32)) return false;   // deflate only, no preset dictionary
    int final_block;
    do {
        final_block = static_cast<int>(br.bits(1));
        const int type = static_cast<int>(br.bits(2));
        if (br.bad) return false;
        if (type == 0) {
            br.align_byte();
            if (br.end - br.p < 4) return false;
            const unsigned len = br.p[0] | (br.p[1] << 8), nlen = br.p[2] | (br.p[3] << 8);
            br.p += 4;
            if ((len ^ 0xffffu) != nlen || static_cast<size_t>(br.end - br.p) < len) return false;
            out.insert(out.end(), br.p, br.p + len);
            br.p += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lengths[320];
            if (type == 1) {
                for (int i = 0; i < 144; i++) lengths[i] = 8;
                for (int i = 144; i < 256; i++) lengths[i] = 9;
                for (int i = 256; i < 280; i++) lengths[i] = 7;
                for (int i = 280; i < 288; i++) lengths[i] = 8;
                lit.build(lengths, 288);
                for (int i = 0; i < 30; i++) lengths[i] = 5;
                dist.build(lengths, 30);
            } else {
                const int hlit = static_cast<int>(br.bits(5)) + 257, hdist = static_cast<int>(br.bits(5)) + 1, hclen = static_cast<int>(br.bits(4)) + 4;
                if (hlit > 286 || hdist > 30) return false;
                uint8_t cl[19] = {0};
                for (int i = 0; i < hclen; i++) cl[cl_order[i]] = static_cast<uint8_t>(br.bits(3));
                Huffman clh;
                if (!clh.build(cl, 19)) return false;
                int i = 0;
                while (i < hlit + hdist) {
                    const int sym = clh.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) { lengths[i++] = static_cast<uint8_t>(sym); continue; }
                    int rep, val = 0;
                    if (sym == 16) { if (i == 0) return false; val = lengths[i - 1]; rep = 3 + static_cast<int>(br.bits(2)); }
                    else if (sym == 17) rep = 3 + static_cast<int>(br.bits(3));
                    else rep = 11 + static_cast<int>(br.bits(7));
                    if (i + rep > hlit + hdist) return false;
                    while (rep--) lengths[i++] = static_cast<uint8_t>(val);
                }
                if (br.bad || !lit.build(lengths, hlit) || !dist.build(lengths + hlit, hdist)) return false;
            }
            for (;;) {
                const int sym = lit.decode(br);
                if (sym < 0) return false;
                if (sym < 256) { out.push_back(static_cast<uint8_t>(sym)); continue; }
                if (sym == 256) break;
                if (sym > 285) return false;
                const int len = len_base[sym - 257] + static_cast<int>(br.bits(len_extra[sym - 257]));
                const int ds = dist.decode(br);
                if (ds < 0 || ds > 29) return false;
                const size_t d = dist_base[ds] + br.bits(dist_extra[ds]);
                if (br.bad || d > out.size()) return false;
                const size_t start = out.size() - d;
                for (int k = 0; k < len; k++) out.push_back(out[start + k]);   // may overlap itself: byte by byte
            }
        } else {
            return false;
        }
    } while (!final_block);
    return !br.bad;
}

// ---- PNG ----------------------------------------------------------------------------------------------------
inline uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
inline int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Undo the scanline filters of one (sub-)image in place: `raw` holds h rows of (1 + stride) bytes.
inline bool unfilter(uint8_t* raw, int h, size_t stride, int bpp /* bytes per complete pixel, >= 1 */)
{
    std::vector<uint8_t> zero(stride, 0);
    const uint8_t* prev = zero.data();
    for (int y = 0; y < h; y++) {
        uint8_t* row = raw + static_cast<size_t>(y) * (stride + 1);
        const int ft = row[0];
        uint8_t* cur = row + 1;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= static_cast<size_t>(bpp) ? cur[i - bpp] : 0, b = prev[i], c = i >= static_cast<size_t>(bpp) ? prev[i - bpp] : 0;
            int v = cur[i];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: return false;
            }
            cur[i] = static_cast<uint8_t>(v);
        }
        prev = cur;
    }
    return true;
}

// bytes (any file contents) -> malloc'ed interleaved 8-bit texels; nullptr on failure
inline unsigned char* decode_memory(const uint8_t* d, size_t n, int* w_out, int* h_out, int* channels_out)
{
    static const uint8_t magic[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (n < 8 + 25 || std::memcmp(d, magic, 8)) return nullptr;
    size_t pos = 8;
    uint32_t W = 0, H = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool have_ihdr = false, done = false;
    while (!done && pos + 12 <= n) {
        const uint32_t len = be32(d + pos);
        const uint8_t* tag = d + pos + 4;
        const uint8_t* body = d + pos + 8;
        if (len > n - pos - 12) return nullptr;
        if (!std::memcmp(tag, "IHDR", 4)) {
            if (len != 13) return nullptr;
            W = be32(body); H = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
            if (body[10] != 0 || body[11] != 0 || interlace > 1 || W == 0 || H == 0 || W > (1u << 24) || H > (1u << 24)) return nullptr;
            have_ihdr = true;
        } else if (!std::memcmp(tag, "PLTE", 4)) plte.assign(body, body + len);
        else if (!std::memcmp(tag, "tRNS", 4)) trns.assign(body, body + len);
        else if (!std::memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(tag, "IEND", 4)) done = true;
        pos += 12 + static_cast<size_t>(len);
    }
    if (!have_ihdr || idat.empty()) return nullptr;
    const int samples = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!samples) return nullptr;
    const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                          (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (!depth_ok || (ctype == 3 && plte.size() < 3)) return nullptr;
    const int bits_pp = samples * depth;
    const int bpp = bits_pp >= 8 ? bits_pp / 8 : 1;

    std::vector<uint8_t> raw;
    if (static_cast<uint64_t>(W) * static_cast<uint64_t>(H) > (1ull << 30)) return nullptr;
    {   // a damaged header must not size the allocation: DEFLATE cannot expand by more than 1032:1
        const size_t expect = (static_cast<size_t>(W) * static_cast<size_t>(bits_pp) / 8 + 2) * static_cast<size_t>(H);
        const size_t ceiling = idat.size() * 1032 + 1024;
        raw.reserve(expect < ceiling ? expect : ceiling);
    }
    if (!inflate(idat.data(), idat.size(), raw)) return nullptr;

    // output channel count (stb_image: req_comp = 0)
    int out_ch = ctype == 3 ? (trns.empty() ? 3 : 4) : samples;
    const bool colour_key = (ctype == 0 && trns.size() >= 2) || (ctype == 2 && trns.size() >= 6);
    if (colour_key) out_ch += 1;
    uint8_t* out = static_cast<uint8_t*>(std::malloc(static_cast<size_t>(W) * H * out_ch));
    if (!out) return nullptr;
    const int grey_scale = depth == 1 ? 255 : depth == 2 ? 85 : depth == 4 ? 17 : 1;
    uint16_t key[3] = {0, 0, 0};
    if (colour_key) for (int k = 0; k < (ctype == 0 ? 1 : 3); k++) key[k] = static_cast<uint16_t>((trns[2 * k] << 8) | trns[2 * k + 1]);

    // one pass = one rectangular sub-image (the whole image when not interlaced)
    static const int ax0[7] = {0, 4, 0, 2, 0, 1, 0}, ay0[7] = {0, 0, 4, 0, 2, 0, 1}, adx[7] = {8, 8, 4, 4, 2, 2, 1}, ady[7] = {8, 8, 8, 4, 4, 2, 2};
    size_t off = 0;
    const int passes = interlace ? 7 : 1;
    for (int p = 0; p < passes; p++) {
        const int x0 = interlace ? ax0[p] : 0, y0 = interlace ? ay0[p] : 0, dx = interlace ? adx[p] : 1, dy = interlace ? ady[p] : 1;
        const uint32_t pw = (W > static_cast<uint32_t>(x0)) ? (W - x0 + dx - 1) / dx : 0, ph = (H > static_cast<uint32_t>(y0)) ? (H - y0 + dy - 1) / dy : 0;
        if (pw == 0 || ph == 0) continue;
        const size_t stride = (static_cast<size_t>(pw) * bits_pp + 7) / 8;
        if (off + (stride + 1) * ph > raw.size()) { std::free(out); return nullptr; }
        uint8_t* sub = raw.data() + off;
        if (!unfilter(sub, static_cast<int>(ph), stride, bpp)) { std::free(out); return nullptr; }
        for (uint32_t yy = 0; yy < ph; yy++) {
            const uint8_t* row = sub + static_cast<size_t>(yy) * (stride + 1) + 1;
            for (uint32_t xx = 0; xx < pw; xx++) {
                uint16_t s[4] = {0, 0, 0, 0};   // samples at file precision
                if (depth == 8) for (int k = 0; k < samples; k++) s[k] = row[static_cast<size_t>(xx) * samples + k];
                else if (depth == 16) for (int k = 0; k < samples; k++) s[k] = static_cast<uint16_t>((row[(static_cast<size_t>(xx) * samples + k) * 2] << 8) | row[(static_cast<size_t>(xx) * samples + k) * 2 + 1]);
                else { const size_t bit = static_cast<size_t>(xx) * depth; s[0] = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1); }
                uint8_t* o = out + (static_cast<size_t>(y0 + yy * dy) * W + (x0 + xx * dx)) * out_ch;
                auto to8 = [&](uint16_t v) -> uint8_t { return depth == 16 ? static_cast<uint8_t>(v >> 8) : static_cast<uint8_t>(v); };
                if (ctype == 3) {
                    const size_t idx = s[0];
                    const bool ok = idx * 3 + 2 < plte.size();
                    o[0] = ok ? plte[idx * 3] : 0; o[1] = ok ? plte[idx * 3 + 1] : 0; o[2] = ok ? plte[idx * 3 + 2] : 0;
                    if (out_ch == 4) o[3] = idx < trns.size() ? trns[idx] : 255;
                } else if (ctype == 0) {
                    o[0] = depth < 8 ? static_cast<uint8_t>(s[0] * grey_scale) : to8(s[0]);
                    if (colour_key) o[1] = s[0] == key[0] ? 0 : 255;
                } else if (ctype == 2) {
                    o[0] = to8(s[0]); o[1] = to8(s[1]); o[2] = to8(s[2]);
                    if (colour_key) o[3] = (s[0] == key[0] && s[1] == key[1] && s[2] == key[2]) ? 0 : 255;
                } else {
                    for (int k = 0; k < samples; k++) o[k] = to8(s[k]);
                }
            }
        }
        off += (stride + 1) * ph;
    }
    *w_out = static_cast<int>(W);
    *h_out = static_cast<int>(H);
    *channels_out = out_ch;
    return out;
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

}  // namespace rtx_png
