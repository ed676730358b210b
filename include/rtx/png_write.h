// png_write.h -- writes an 8-bit RGB / RGBA image as a PNG file (presentation).
//
// The reference's last step is glfwSwapBuffers (main.cpp:189): the frame goes to a window. A head-less replacement can
// only hand the frame to the host (rtx_read_pixels) -- this is the smallest way to look at it. No compression: the image
// data is a zlib stream of stored DEFLATE blocks (RFC 1950 / 1951 section 3.2.4), every scanline with filter type 0,
// so the file is valid for any PNG reader (include/rtx/png_decode.h and Pillow read it back in tests/test_png_decode.py).
#pragma once
#include <cstdint>
#include <cstdio>
#include <vector>

namespace rtx_png {

inline uint32_t crc32_update(uint32_t c, const uint8_t* p, size_t n)   // PNG annex D, bitwise
{
    for (size_t i = 0; i < n; ++i) {
        c ^= p[i];
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
    }
    return c;
}

// pixels: h rows of w * channels bytes; row 0 = TOP row unless bottom_up (rtx_read_pixels returns row 0 = bottom row,
// like gl_FragCoord). channels: 3 (RGB) or 4 (RGBA). Returns false on I/O failure or bad arguments.
inline bool write_file(const char* path, const unsigned char* pixels, int w, int h, int channels, bool bottom_up = false)
{
    if (!pixels || w <= 0 || h <= 0 || (channels != 3 && channels != 4)) return false;
    const size_t stride = static_cast<size_t>(w) * static_cast<size_t>(channels);
    std::vector<uint8_t> raw;
    raw.reserve((stride + 1) * static_cast<size_t>(h));
    for (int y = 0; y < h; ++y) {
        const unsigned char* row = pixels + stride * static_cast<size_t>(bottom_up ? h - 1 - y : y);
        raw.push_back(0);   // filter type: none
        raw.insert(raw.end(), row, row + stride);
    }
    std::vector<uint8_t> z;   // zlib stream: header, stored blocks of <= 65535 bytes, Adler-32
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t pos = 0; pos < raw.size() || pos == 0;) {
        const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
        const bool last = pos + n >= raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back(static_cast<uint8_t>(n & 255));
        z.push_back(static_cast<uint8_t>(n >> 8));
        z.push_back(static_cast<uint8_t>(~n & 255));
        z.push_back(static_cast<uint8_t>((~n >> 8) & 255));
        for (size_t i = 0; i < n; ++i) {
            a = (a + raw[pos + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        z.insert(z.end(), raw.begin() + static_cast<std::ptrdiff_t>(pos), raw.begin() + static_cast<std::ptrdiff_t>(pos + n));
        pos += n;
        if (last) break;
    }
    const uint32_t adler = (b << 16) | a;
    for (int k = 3; k >= 0; --k) z.push_back(static_cast<uint8_t>(adler >> (8 * k)));

    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    auto be32 = [](uint8_t* d, uint32_t v) { d[0] = static_cast<uint8_t>(v >> 24); d[1] = static_cast<uint8_t>(v >> 16); d[2] = static_cast<uint8_t>(v >> 8); d[3] = static_cast<uint8_t>(v); };
    auto chunk = [&](const char* tag, const uint8_t* body, size_t n) {
        uint8_t head[8], tail[4];
        be32(head, static_cast<uint32_t>(n));
        for (int k = 0; k < 4; ++k) head[4 + k] = static_cast<uint8_t>(tag[k]);
        uint32_t c = crc32_update(0xFFFFFFFFu, head + 4, 4);
        if (n) c = crc32_update(c, body, n);
        be32(tail, c ^ 0xFFFFFFFFu);
        return std::fwrite(head, 1, 8, f) == 8 && (n == 0 || std::fwrite(body, 1, n, f) == n) && std::fwrite(tail, 1, 4, f) == 4;
    };
    static const uint8_t magic[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    uint8_t ihdr[13];
    be32(ihdr, static_cast<uint32_t>(w));
    be32(ihdr + 4, static_cast<uint32_t>(h));
    ihdr[8] = 8;
    ihdr[9] = channels == 3 ? 2 : 6;
    ihdr[10] = ihdr[11] = ihdr[12] = 0;
    bool ok = std::fwrite(magic, 1, 8, f) == 8 && chunk("IHDR", ihdr, 13) && chunk("IDAT", z.data(), z.size()) && chunk("IEND", nullptr, 0);
    ok = (std::fclose(f) == 0) && ok;
    return ok;
}

}  // namespace rtx_png
