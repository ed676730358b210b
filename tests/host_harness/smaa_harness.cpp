// smaa_harness.cpp -- TEST INFRASTRUCTURE: the product's SMAA arithmetic (raytracing_opengl_amd/csrc/smaa_device.h) compiled for the
// HOST and run densely over a frame, so that it can be compared byte for byte with the oracle without a GPU
// (tests/test_smaa_host.py, -m "not gpu"). Not a product path: the product runs these functions only inside the HIP kernels
// (smaa_kernel.hip: LDS luma tile, edge list, sparse passes -- that plumbing is what the -m gpu tests check).
#include <cstdint>
#include <vector>

#include "smaa_device.h"

#include <vector>

static int use_planes = 1;   // 0: every orthogonal search takes the per-step loop (the form the planes' step count is checked against)
extern "C" void harness_smaa_use_planes(int on) { use_planes = on; }
static int use_roles = 0;    // 1: a pixel's weights from the independent parts + the selection rule, the diagonal part once per pair of diagonals --
                             // the composition smaa_weights_roles_kernel runs on four waves (smaa_kernel.hip) -- instead of weights()
extern "C" void harness_smaa_use_roles(int on) { use_roles = on; }
template <class B>
static uint32_t weights_by_roles(const B& b, const smaa::Preset& P, int x, int y)
{
    const float X = (float)x, Y = (float)y;
    const smaa::F2 e = b.own_edges(x, y);
    smaa::F2 d1{0.0f, 0.0f}, d2{0.0f, 0.0f}, north{0.0f, 0.0f}, west{0.0f, 0.0f};
    if (b.has_diag_part(e)) d1 = b.part_diag(X, Y, e, 1u);
    if (b.has_diag_part(e)) d2 = b.part_diag(X, Y, e, 2u);
    if (e.y > 0.0f) north = b.part_north(X, Y);
    if (e.x > 0.0f) west = b.part_west(X, Y);
    return B::combine(e, P.max_steps_diag > 0, smaa::F2{d1.x + d2.x, d1.y + d2.y}, north, west);
}
extern "C" int harness_smaa(const uint32_t* color, int w, int h, int preset, const uint16_t* area, const uint8_t* search, uint16_t* edges,
                            uint32_t* blend, uint32_t* screen)
{
    if (!color || !area || !search || !edges || !blend || !screen || w <= 0 || h <= 0 || preset < 0 || preset > 3) return -1;
    const smaa::Preset P = smaa::preset_of(preset);
    auto luma = [&](int x, int y) {
        x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
        y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
        return smaa::luma_of(color[(size_t)y * w + x]);
    };
#pragma omp parallel for
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            edges[(size_t)y * w + x] = (uint16_t)smaa::edge_from_lumas(P.threshold, luma(x, y), luma(x - 1, y), luma(x, y - 1), luma(x + 1, y), luma(x, y + 1),
                                                                       luma(x - 2, y), luma(x, y - 2));
    const smaa::Views V{w, h, color, edges, blend, area, search};
    // the bit planes of the edge texture, as the dense HIP kernel writes them (smaa_kernel.hip): the orthogonal searches count their steps
    // on these (smaa::SearchPlanes), exactly like the weight kernel
    const int pw = smaa::SearchPlanes::plane_words(w);
    std::vector<uint64_t> prow((size_t)pw * h, 0ull);
    std::vector<uint16_t> pcol((size_t)((h + 7) / 8) * w, 0);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint16_t e = edges[(size_t)y * w + x];
            const uint64_t two = ((e & 0x00ffu) ? 1ull : 0ull) | ((e & 0xff00u) ? 2ull : 0ull);
            prow[(size_t)y * pw + (x >> 5)] |= two << (2 * (x & 31));
            pcol[(size_t)(y >> 3) * w + x] |= (uint16_t)(two << (2 * (y & 7)));
        }
    const smaa::SearchPlanes planes{use_planes ? prow.data() : nullptr, pcol.data(), w, h};
    if (use_planes) {
        // the HIP kernels' form: single edge texels from the row plane too (PlaneTex), a weight texture that is NEVER cleared -- filled
        // with garbage here wherever the frame has no edge pixel -- and pass 3 looking at weights only where the plane has an edge
        const smaa::PlaneTex src{prow.data(), pw};
        const smaa::BlendT<smaa::PlaneTex> B{V, P, planes, src};
#pragma omp parallel for schedule(dynamic, 4)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
                blend[(size_t)y * w + x] = edges[(size_t)y * w + x] ? (use_roles ? weights_by_roles(B, P, x, y) : B.weights(x, y)) : (0x9e3779b9u * (uint32_t)(y * w + x + 1)) | 0x01010101u;
#pragma omp parallel for
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                uint32_t out;
                screen[(size_t)y * w + x] = smaa::neighborhood<true>(V, x, y, out, &src) ? out : color[(size_t)y * w + x];
            }
        for (int y = 0; y < h; y++)                      // what a read-back of the weight texture returns: stale texels masked out
            for (int x = 0; x < w; x++)
                if (!edges[(size_t)y * w + x]) blend[(size_t)y * w + x] = 0u;
        return 0;
    }
    const smaa::TexEdges src{edges, w};
    const smaa::Blend B{V, P, planes, src};
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) blend[(size_t)y * w + x] = edges[(size_t)y * w + x] ? (use_roles ? weights_by_roles(B, P, x, y) : B.weights(x, y)) : 0u;
#pragma omp parallel for
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t out;
            screen[(size_t)y * w + x] = smaa::neighborhood(V, x, y, out) ? out : color[(size_t)y * w + x];
        }
    return 0;
}

// exhaustive check of smaa::unorm8 (divide-free) against byte / 255.0f
extern "C" int harness_smaa_unorm8_mismatches()
{
    int bad = 0;
    for (uint32_t b = 0; b < 256; b++) {
        volatile float ref = (float)b / 255.0f;
        if (smaa::unorm8(b) != ref) bad++;
    }
    return bad;
}
