// trgl_device.h — structs shared by the host side (trgl_api.cpp) and the gfx950 kernels.
//
// Data layout in HBM (one context):
//   fb      uint8  [H][W][bpp]      TGAImage layout, row y=0 first (tgaimage.cpp:32-39)
//   zb      double [H][W]           the reference's global zbuffer (our_gl.cpp:15,72-74)
//   recs    TriRec [N]              one 128-B line per submitted triangle, written by setup
//   recs_w  TriW   [N]              1/w of the vertices, for draws whose fragments need perspective-correct barycentrics
//   cnt/off uint32 [N]              tiles overlapped per triangle and its exclusive scan
//   keys/vals/bmask [P] x2          (tile id, triangle id, 4x4 mask of the tile's blocks the bbox reaches) triples, ping-pong
//                                   for the radix passes
//   tile_start/tile_end uint32[T]   per-tile slice of the sorted pair list
#pragma once
#include <stdint.h>
#include "../../include/trgl.h"

#define TRGL_TILE      32          // tile is TILE x TILE pixels = 4 x 4 BLOCKS of 8 x 8 pixels; one wavefront owns one block
#define TRGL_TILE_LOG2 5
#define TRGL_TILE_PIX  (TRGL_TILE * TRGL_TILE)
#define TRGL_BLOCK_LOG2 3          // a block is 8 x 8 pixels = the 64 lanes of a wavefront (lane = 8 * row + column)
#define TRGL_MAX_DRAWS 64          // draws per flush (each with its own shader kind + uniforms)

// Per-triangle setup record: everything the pixel loop needs, hoisted exactly as SURVEY §8(a) A4/A6
// allows (same operations on the same operands as our_gl.cpp:77-86,168-170, so bit-identical).
// The whole record is what a visit of k_raster takes as wave-uniform constants (two scalar loads of 64 B); the per-block cull test
// reads chunks (16 B) 0-4 and 7 per candidate lane.
struct alignas(128) TriRec {
    double ax, ay;            // screen[0]                       (our_gl.cpp:117-121)
    double s0x, s0y;          // C.x - A.x, B.x - A.x            (our_gl.cpp:78)
    double s1x, s1y;          // C.y - A.y, B.y - A.y            (our_gl.cpp:79)
    double c0;                // depth plane of the early depth test: every covered pixel of the triangle has
                              //   z > c0 + (ax - x) g1 + (ay - y) g2   (k_setup; c0 = -inf, g = 0: no test)
    double uz;                // s0x*s1y - s0y*s1x = cross().z   (our_gl.cpp:80, geometry.h:147)
    double g1, g2;
    double ruz;               // RN(1/uz) when the triangle is "well scaled" (see setup), else 0:
                              // lets the pixel loop divide by uz with FMAs, bit-identically
    double z0, z1, z2;        // NDC z of the three vertices     (our_gl.cpp:156-158)
    uint16_t bx0, by0, bx1, by1;   // clamped pixel bbox, inclusive (our_gl.cpp:130-133)
    uint32_t color;           // FLAT packed BGRA (GOURAUD reads its base colour through `dl`)
    uint32_t dl;              // bit 31: not "well scaled" (literal divisions, see k_setup) | draw index << 24 | triangle index inside its draw (< 2^24)
};
static_assert(sizeof(TriRec) == 128, "TriRec must be one 128-B line");
#define TRGL_DRAW_MAX_TRIS (1u << 24)   // triangles per DrawDesc; trgl_draw splits larger submissions
#define TRGL_FLUSH_MAX_TRIS ((1u << 25) - 1u)  // triangles per flush: k_raster addresses a record (and the one behind the last) as base + (index << 7) with a 32-bit scalar offset
// A pair's triangle word: index in the flush (25 bits, TRGL_FLUSH_MAX_TRIS) | a 7-bit lower bound of the triangle's depths,
// zq: every covered pixel has z >= -1 + zq / 64 (k_setup; 0 = no bound).
#define TRGL_VAL_TRI(v)   ((v) & 0x1ffffffu)
#define TRGL_VAL_ZQ(v)    ((v) >> 25)
#define TRGL_DL_LITERAL   0x80000000u
#define TRGL_DL_DRAW(dl)  (((dl) >> 24) & (TRGL_MAX_DRAWS - 1))
#define TRGL_DL_LOCAL(dl) ((dl) & 0xffffffu)
#define TRGL_DL_ID(dl)    ((dl) & 0x3fffffffu)          // draw << 24 | local: what the visibility buffer holds (~0u = no owner)

// 1/w of the three vertices, |w|>1e-12 ? 1/w : 0 (our_gl.cpp:168-170): read only where perspective-correct barycentrics are
// needed (GOURAUD, PHONG, EYE fragments), so it lives beside the record and is written only for draws of those kinds.
struct alignas(32) TriW { double iw0, iw1, iw2, pad; };

struct DevTexture {
    const uint8_t* data;
    int32_t w, h, bpp, pad;
};

struct DrawDesc {             // one trgl_draw() call, resident on the device for the flush
    const double*   clip;     // [n][12]
    const double*   vary;     // [n][K] or null
    const uint32_t* colors;   // [n] or null
    uint32_t        n;
    uint32_t        first;    // global index of this draw's first triangle within the flush
    int32_t         kind;
    int32_t         K;
    trgl_uniforms   u;
};

// Device-side mirror of the reference's counters (our_gl.cpp:18-22).  z range is kept as
// order-preserving uint64 keys so atomicMin/atomicMax work on it.
struct DevStats {
    unsigned long long fragments;
    unsigned long long zmin_key, zmax_key;
    int32_t min_x, min_y, max_x, max_y;
    unsigned long long pairs_total;      // written by the scan spine (implementation traffic)
    unsigned long long literal_tris;     // triangles of the flush in flight that are not "well scaled" (k_setup adds, k_fold_stats clears); copied to the
                                         // host together with pairs_total: a flush without any runs the raster kernel that has no literal path
    unsigned long long large_tris;       // ... whose bbox is at least 64 pixels wide or high (k_setup adds, k_fold_stats clears): a flush with any lets the list steps of
                                         // k_raster drop entries by the depth bound that rides in the pair (TRGL_VAL_ZQ)
    // std::min/std::max keep the FIRST of two equal values (our_gl.cpp:197-198), and +0.0 == -0.0:
    // when the z range ends in a zero its sign is that of the first zero written, in the reference's
    // order (triangle, x, y).  Keys = tri<<32 | x<<16 | y of the first +0 / -0 fragment of this flush.
    unsigned long long zero_pos_key, zero_neg_key;
    uint32_t zero_locked, zero_sign;     // set by k_fold_stats once any zero has been written
    unsigned long long dbg[16];           // diagnostic builds only (-DTRGL_DEBUG_COUNTERS): work counters of k_raster
};
#define TRGL_ZERO_KEY_EMPTY 0xffffffffffffffffull

struct FrameParams;
#ifdef __HIPCC__
#define TRGL_HD __host__ __device__ __forceinline__
#else
#define TRGL_HD inline
#endif
struct FrameParams {
    uint8_t* fb;
    double*  zb;
    uint32_t* idbuf;                  // [H][W] draw << 24 | index in the draw of the triangle that owns the pixel (PHONG / EYE flushes), ~0u = none
    int32_t  W, H, bpp;
    int32_t  tiles_x, tiles_y;
    int32_t  strip_y0, strip_y1;      // rows this context owns
    int32_t  strip_ty0, strip_ty1;    // tile rows intersecting the strip: [ty0, ty1)
    int32_t  il_tiles, il_world, il_rank;   // interleaved ownership (trgl_set_interleave): bands of il_tiles tile rows dealt round-robin to
                                      // il_world contexts, this one takes band number == il_rank (mod il_world); il_tiles = 0: the strip above
    uint32_t n_tris;                  // triangles of the flush (diagnostic builds check list entries against it)
    int32_t  init_from_clear;         // 1: tiles start from the clear values, not from HBM
    int32_t  zq_cull;                 // 1: the flush holds large triangles - k_raster's list steps compare TRGL_VAL_ZQ with the block's largest depth
    uint32_t clear_color;             // packed BGRA
    double   clear_z;
    double   vp[8];                   // rows 0 and 1 of the Viewport matrix (our_gl.cpp:117-121)
};

// ---- which tile rows a context owns (one strip, or interleaved bands) ----------------------------------------------
TRGL_HD bool tile_row_owned(const FrameParams& fp, int ty) {
    if (fp.il_tiles == 0) return ty >= fp.strip_ty0 && ty < fp.strip_ty1;
    return (ty / fp.il_tiles) % fp.il_world == fp.il_rank;
}
// interleaved ownership: number of owned tile rows below row r
TRGL_HD int il_owned_below(const FrameParams& fp, int r) {
    const int period = fp.il_tiles * fp.il_world, q = r / period, m = r - q * period - fp.il_rank * fp.il_tiles;
    return q * fp.il_tiles + (m < 0 ? 0 : (m > fp.il_tiles ? fp.il_tiles : m));
}
// ... and the j-th owned tile row counted from row ty0 (j = 0: the first owned row >= ty0)
TRGL_HD int il_nth_owned_from(const FrameParams& fp, int ty0, int j) {
    const int c = il_owned_below(fp, ty0) + j;
    return (c / fp.il_tiles) * (fp.il_tiles * fp.il_world) + fp.il_rank * fp.il_tiles + (c % fp.il_tiles);
}
