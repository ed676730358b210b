// rt_scene_dev.h -- the tracer's device-side scene tables ("DevScene" blob).
//
// The boundary receives nine std140 uniform blocks (rt.frag:155-230, byte layouts in
// include/rtx/scene.h). They are re-packed on the host, once per block update, into ONE
// contiguous blob that the kernel reads with wave-uniform indices (scalar loads, or LDS when
// staged): geometry records first (scanned by every ray), 64-byte materials behind them
// (fetched once per hit). Every derived field is a pure function of one primitive and is
// computed with the same float operations the shader would execute per ray (quat_inv, r*r,
// 4*R*R, ...), so hoisting it changes no result bit.
//
// All records are multiples of 16 bytes and the blob base is 16-byte aligned, so any record
// field group can be fetched with s_load_dwordx4 / ds_read_b128.
#pragma once
#include <stdint.h>

namespace rtdev {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct alignas(16) f4 { float x, y, z, w; };

enum PrimType { TYPE_SPHERE = 0, TYPE_PLANE = 1, TYPE_SURFACE = 2, TYPE_BOX = 3, TYPE_TORUS = 4, TYPE_RING = 5, TYPE_POINT_LIGHT = 6 };

struct alignas(16) DevMaterial {  // byte-identical to std140 rt_material (rt.frag:24-34)
    float color[3]; float _p0;
    float absorb[3];
    float diffuse;
    float reflection;
    float refraction;
    int32_t specular;
    float kd;
    float ks;
    float _p1[3];
};

struct alignas(16) DevSphere {
    f4 geom;           // centre xyz, w = r*r   (rt.frag:346)
    int32_t hollow;    // closest-hit only (trap T8)
    int32_t texture;   // textureNum
    float radius;
    int32_t _p;
    f4 quat;           // rotates the NORMAL for texturing (rt.frag:319-322)
};
struct alignas(16) DevPlane {
    f4 normal;         // xyz (not normalised, rt.frag:357)
    f4 pos;
};
struct alignas(16) DevSurface {
    f4 quat;
    f4 pos_a;          // pos xyz, a
    f4 bcde;           // b, c, d, e
    f4 f_vmin;         // f, v_min xyz (world-space clip box, trap T6)
    f4 vmax;           // v_max xyz, w = int bits: 1 if quat is the identity (any zero signs)
    f4 qinv;           // quat_inv(quat)
};
#ifndef RT_QUADRIC_FAR
#define RT_QUADRIC_FAR 64.0   /* a quadric bound that rests on the surface's own extent holds for origins up to this far from its centre (rt_pack.h pack_scene) */
#endif
struct alignas(16) DevSurfaceCull {  // first-level record of a quadric (surface_cull)
    f4 bound;          // cull sphere of the CLIPPED surface: centre xyz (world), w = radius^2 (inflated); w < 0: unbounded. It may rest on the
                       // surface's own extent (the FATTENED surface the reference's float arithmetic sees, rt_pack.h) and then holds for origins
                       // within RT_QUADRIC_FAR of the centre only
    f4 sym0;           // symmetric M = R^T diag(a,b,c) R : m00, m01, m02, m11
    f4 sym1;           // m12, m22, |p2| margin, w = radius^2 about the same centre of a bound that holds for origins at ANY distance (the sphere
                       // of a closed clip box: a hit lies in the box whatever the arithmetic does) -- what far origins and the candidate tables
                       // (pencils, slab tables, group culls) use; NaN: none (a clip box open along some axis)
};
struct alignas(16) DevBox {
    f4 quat;
    f4 pos;            // xyz, w = int bits: 1 if quat is the identity
    f4 form_tex;       // half extents xyz, w = textureNum as float bits (int)
    f4 qinv;
};
#ifndef RT_TORUS_HULL_MARGIN
#define RT_TORUS_HULL_MARGIN 2.5e-4   /* how far outside the torus' convex hull an origin must lie for the hull cull (rt_device.h torus_local_cull) */
#endif
#ifndef RT_TORUS_IM_NOISE
#define RT_TORUS_IM_NOISE 0.1         /* the largest |Im| of a complex root pair the reference's solver may take for real: measured 0.057 from 30 .. 60 units out
                                         (tools/cull_audit.py torus_margin on the sized-torus scenes, profiles/r06*); every torus bound is inflated for it (rt_pack.h) */
#endif
/* the torus sizes the cull premises were audited on (tools/cull_audit.py; rt_pack.h): others are never culled */
#define RT_TORUS_CULL_R_MIN 0.25f
#define RT_TORUS_CULL_R_MAX 2.5f
#define RT_TORUS_CULL_TUBE_MIN 0.08f
#define RT_TORUS_CULL_TUBE_MAX 2.0f
struct alignas(16) DevTorus {
    f4 quat;
    f4 pos;            // xyz, w = int bits: 1 if quat is the identity
    f4 radii;          // R, r, R*R, r*r
    f4 k;              // x = 4*R*R, y = z = puck radius^2 = world cull-sphere radius^2 ((R+r) inflated), w = hole radius^2 ((R-r) deflated, 0 = none)
    f4 qinv;
    f4 cull;           // x = puck half height (r inflated), y = (r + RT_TORUS_HULL_MARGIN)^2, z = |R| (convex-hull cull), w = r + RT_TORUS_HULL_MARGIN (start cull)
};
struct alignas(16) DevRing {
    f4 quat;
    f4 pos_tex;        // pos xyz, w = textureNum (int bits)
    f4 radii;          // r1, r2 (squared radii, trap T7), r2 - r1, w = cull radius^2 (r2 inflated)
    f4 normal;         // rotate(quat_inv(quat), (0,0,-1))  (rt.frag:391-394), w = int bits: 1 if quat is the identity
};
struct alignas(16) DevLightPoint {
    f4 pos_r2;         // xyz, w = radius*radius (light sphere, closest-hit only)
    f4 color_intensity;
    f4 atten;          // linear_k, quadratic_k, radius, unused
};
struct alignas(16) DevLightDirect {
    f4 direction;      // xyz as given (not normalised)
    f4 color_intensity;
    f4 dir_n;          // normalize(-direction): the light vector of calcShade (rt.frag:700-703), hoisted -- the same
                       // IEEE sqrt and divisions on the same operands, done once on the host instead of per shaded lane
};

struct alignas(16) DevSceneHeader {
    int32_t n_sphere, n_plane, n_surface, n_box;
    int32_t n_torus, n_ring, n_light_point, n_light_direct;
    int32_t iterations, canvas_w, canvas_h, total_bytes;
    f4 cam_quat;
    f4 cam_pos;
    f4 ambient;         // AMBIENT_COLOR after the %f text round trip (trap T9)
    f4 shadow_ambient;  // SHADOW_AMBIENT, same
    // byte offsets of the arrays from the blob base
    uint32_t off_sphere, off_plane, off_surface, off_box;
    uint32_t off_torus, off_ring, off_light_point, off_light_direct;
    uint32_t off_mat[8];  // materials per PrimType 0..5 (6,7 unused)
    // first-level ("cull") arrays: 16-byte records scanned 4 at a time with one batched scalar load.
    // Each is padded with zero records to a multiple of 4 entries (surf_cull: of 2).
    uint32_t off_sph_geom;     // f4 per sphere: centre, r*r  (the sphere test itself)
    uint32_t off_sph_hollow;   // uint32 bit per sphere (bit i&31 of word i>>5)
    uint32_t off_surf_cull;    // DevSurfaceCull per quadric
    uint32_t off_torus_bound;  // f4 per torus: centre, inflated bounding radius^2
    uint32_t off_ring_bound;   // f4 per ring:  centre, inflated outer radius^2
    uint32_t cam_ident;        // 1 if cam_quat is the identity (any zero signs): getRayDir's rotation is then v + 0.0f
    // second-level cull arrays for long tables: one f4 per group of RT_GROUP consecutive quadrics / tori = a sphere around the group's
    // first-level bounds (centre, radius^2; w < 0 or inf: the group has an unbounded member and is never culled)
    uint32_t off_surf_group;
    uint32_t off_torus_group;
    // ray pencils (DevPencil below): n_pencil = 0 when the scene has none. The masks themselves are built on the device (rt_kernel.hip
    // pencil_build_kernel) into a buffer of their own; pencil_stride = mask words per cell (quadric words first, then torus words)
    uint32_t n_pencil, off_pencil, pencil_stride, pencil_mask_words;
    // rays outside every pencil (mirror / refracted rays, shadow rays of lights without one): slab tables (DevSlabs below; 0 = none) and
    // the direction table of the quadrics' degenerate branch (record n_pencil of the pencil array; 0xffffffff = none)
    uint32_t off_slabs, pencil_dir, _pad[2];
};

// ---- ray pencils: third-level cull for long tables ---------------------------------------------
// Most rays of a frame belong to one of a few PENCILS: camera rays all start at the eye, shadow rays towards a point light all end at
// the light, shadow rays towards a directional light are all parallel. A pencil's rays are indexed by two numbers (a direction from the
// apex: cube-map cell; or a position in the plane across the common direction: grid cell), and for every cell the builder records which
// quadrics / tori a ray of that cell could possibly need -- one bit per primitive, conservative (bit clear = the primitive's own
// first-level cull would provably reject every ray of the cell). A scan then visits the set bits of the wave's OR of its lanes' cells
// instead of walking the whole table. Rays outside any pencil (mirror / refracted rays) keep the two-level scan.
enum { RT_MAX_PENCILS = 8, RT_PENCIL_APEX_RES = 64, RT_PENCIL_PAR_RES = 128, RT_PENCIL_DIR_RES = 32, RT_PENCIL_MAX_PRIMS = 128, RT_PENCIL_MIN_PRIMS = 16 };
// RT_PENCIL_DIRECTION is not a pencil of rays but the same kind of table for ANY ray, indexed by its direction alone: bit i set = quadric i
// might take its degenerate branch (trap T4: |p2| < 1e-6, p2 = d^T M d) for a direction of the cell -- the part of "can this ray need
// quadric i" that no table over positions can answer, because a quadric on that branch ignores its clip box.
enum { RT_PENCIL_OFF = 0, RT_PENCIL_APEX = 1, RT_PENCIL_PARALLEL = 2, RT_PENCIL_DIRECTION = 3 };
struct alignas(16) DevPencil {
    f4 a;          // APEX: the common point, w = 0: the rays start there (camera), 1: they run towards it (point light); PARALLEL: the common (unit) direction
    f4 e1;         // PARALLEL: first in-plane axis xyz, w = coordinate of the low edge of cell 0
    f4 e2;         // PARALLEL: second axis, w likewise
    f4 grid;       // PARALLEL: x, y = cells per unit length along e1, e2; z, w = cell sizes
    int32_t kind;  // RT_PENCIL_*
    int32_t res;   // cells per cube-face edge (APEX) / per axis (PARALLEL)
    uint32_t cells;     // number of cells; cell number `cells` is the all-ones cell for rays the pencil cannot vouch for
    uint32_t mask_off;  // first dword of cell 0 in the mask buffer
};

// ---- slab tables: candidate masks for rays that belong to no pencil ------------------------------
// The box around all bounded quadrics / tori is cut into RT_SLABS slabs along each axis; T[axis][0][i] = the primitives whose (padded) bound
// reaches into slab i, T[axis][l][i] = the OR of 2^l consecutive slabs from i on (a sparse table: the OR over any slab range is two
// entries). A piece of a ray whose end points lie in slabs [ax..bx] x [ay..by] x [az..bz] can only meet primitives in
// OR(x range) & OR(y range) & OR(z range); a ray is cut into RT_SLAB_SEGMENTS pieces between its entry into the box and its exit (or its
// length limit) and the pieces' masks are ORed, together with the primitives that have no usable bound (`always`) and the quadrics the
// direction table names. Built on the host with the scene (a few hundred interval insertions); read with vector loads.
enum { RT_SLABS = 64, RT_SLAB_LEVELS = 7, RT_SLAB_MAX_WORDS = 4, RT_SLAB_SEGMENTS = 4 };
struct alignas(16) DevSlabs {
    f4 lo, hi;             // the box (padded)
    f4 inv;                // slabs per unit length along x, y, z
    uint32_t always[4];    // mask words (quadric words first, then torus words, like a pencil cell) of the primitives in every ray's mask
    uint32_t valid[4];     // every primitive that exists: the mask of a ray the tables cannot vouch for
    uint32_t table_off;    // byte offset in the blob of T[axis][level][slab][word]
    uint32_t _pad[3];
};

// ---- textures --------------------------------------------------------------------------------
// Texels are stored as RGBA8 (one dword per texel) whatever the source channel count, so a tap
// is one aligned 4-byte load; RGB sources get alpha 255 (= 1.0 exactly), GL_RED gets (r,0,0,255).
enum { RT_GROUP = 8 };   // primitives per second-level cull group (consecutive indices: scan order is untouched)

enum { TEX_SPHERE_1 = 0, TEX_SPHERE_2, TEX_SPHERE_3, TEX_SPHERE_4, TEX_RING, TEX_BOX, TEX_SLOTS };
enum { MAX_MIPS = 15 };

struct DevTexture {
    const uint32_t* texels;  // level 0 at offset 0; nullptr = unbound sampler (samples black, alpha 1)
    int32_t width, height;
    int32_t wrap;            // 0 REPEAT, 1 CLAMP_TO_EDGE
    int32_t levels;
    float fwidth, fheight;   // (float)width, (float)height: kept as scalars instead of per-lane conversions
    uint32_t level_off[MAX_MIPS];  // dword offset of each mip level
};
struct DevCubemap {
    const uint32_t* texels;  // level 0: 6 faces back to back (+X,-X,+Y,-Y,+Z,-Z), each size*size dwords; level L (load_cubemap(faces, true))
                             // follows level L-1: 6 faces of max(1, size>>L)^2 dwords each
    int32_t size;
    int32_t face_mask;       // bit f set = face present (a missing face samples black)
    float fsize;             // (float)size
    int32_t levels;          // 1 = no mip chain (the reference's default genMipmap = false, or RTX_OPT_TEXTURE_LOD = 0)
    uint32_t level_off[MAX_MIPS];  // dword offset of each level's six faces (round 6, ADVICE r5: the fetch used to add the levels up per lane)
};

}  // namespace rtdev
