// kernels_bin.hip — per-triangle setup and the order-preserving binning prepass (gfx950).
//
//   setup      : our_gl.cpp:89-141 for every triangle -> TriRec + number of tiles overlapped
//   scan       : exclusive scan (counts -> pair offsets; radix histograms -> scatter bases)
//   expand     : (tile id, triangle id) pairs in triangle order
//   radix pass : stable LSD counting sort of the pairs by tile id, so every tile's slice lists its
//                triangles in SUBMISSION ORDER (z ties keep the earlier triangle, our_gl.cpp:165;
//                fragments_drawn / z-range are order dependent, our_gl.cpp:194-198)
//   bounds     : per-tile [start,end) in the sorted pair list
//
// All integer / fp64 IEEE work; built with -ffp-contract=off so no product is fused into a sum.
#include <hip/hip_runtime.h>
#include "trgl_device.h"
#include "launch.h"

namespace {

__device__ __forceinline__ int x86_cvttsd2si(double d) {
    // (int)double exactly as the reference's x86-64 build executes it: out of range -> INT_MIN
    if (!(d > -2147483649.0 && d < 2147483648.0)) return INT_MIN;
    return (int)d;
}
__device__ __forceinline__ double dmin3(double a, double b, double c) { double m = a; if (b < m) m = b; if (c < m) m = c; return m; }
__device__ __forceinline__ double dmax3(double a, double b, double c) { double m = a; if (m < b) m = b; if (m < c) m = c; return m; }
__device__ __forceinline__ double dot4(const double* m, const double* v) {
    double sum = 0;                               // geometry.h:122-127: left to right from 0
    sum += m[0] * v[0]; sum += m[1] * v[1]; sum += m[2] * v[2]; sum += m[3] * v[3];
    return sum;
}

__device__ __forceinline__ int wave_min_i(int v) {
    for (int o = 32; o; o >>= 1) { int t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
    for (int o = 32; o; o >>= 1) { int t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}

// ---------------------------------------------------------------------------------------------
// setup: one thread per triangle of one draw.  Follows our_gl.cpp:89-141 line by line.
// HBM traffic is staged through LDS so that both the 96-B/triangle clip stream and the 128-B/triangle
// record stream move as 16 B per lane, fully coalesced (a lane reading its own 96-B triangle straight
// from HBM touches 64 different lines per load instruction).
// ---------------------------------------------------------------------------------------------
constexpr int SETUP_THREADS = 256;

__global__ __launch_bounds__(SETUP_THREADS) void k_setup(FrameParams fp, DrawDesc d, DrawDesc* __restrict__ draws_out, int draw_idx,
                                                         TriRec* __restrict__ recs, TriW* __restrict__ recs_w, uint32_t* __restrict__ cnt,
                                                         uint2* __restrict__ tilebox, DevStats* __restrict__ stats,
                                                         uint32_t* __restrict__ blk_sums, uint32_t blk_base) {
    __shared__ __attribute__((aligned(16))) double s_buf[SETUP_THREADS * 16];    // 32 KB: in [256][12], then out [256][16]
    // The draw's descriptor arrives as a kernel argument (scalar loads from the argument segment; no copy command on the stream before
    // the flush's first kernel) and is left in device memory for the kernels behind this one, which read uniforms and arrays through it.
    if (blockIdx.x == 0 && threadIdx.x == 0) draws_out[draw_idx] = d;
    const uint32_t b0 = blockIdx.x * SETUP_THREADS;
    const uint32_t nb = min((uint32_t)SETUP_THREADS, d.n - b0);
    const uint32_t tid = threadIdx.x;
    const uint32_t i = b0 + tid;
    const bool in_range = tid < nb;

    // ---- clip stream in: nb*96 contiguous bytes ------------------------------------------------
    const double* src = d.clip + (size_t)b0 * 12;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const double2* s2 = reinterpret_cast<const double2*>(src);
        double2* l2 = reinterpret_cast<double2*>(s_buf);
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        for (uint32_t k = tid; k < nb * 6; k += SETUP_THREADS)
            *reinterpret_cast<f64x2*>(&l2[k]) = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(&s2[k]));
    } else {
        for (uint32_t k = tid; k < nb * 12; k += SETUP_THREADS) s_buf[k] = src[k];
    }
    __syncthreads();
    double v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = in_range ? s_buf[tid * 12 + k] : 0.0;
    __syncthreads();

    int bx0 = INT_MAX, by0 = INT_MAX, bx1 = INT_MIN, by1 = INT_MIN;   // contribution to bbox stats
    uint32_t ntiles = 0;
    bool large = false;
    uint2 tb = make_uint2(0, 0);
    TriRec r;
    {
        uint4* z4 = reinterpret_cast<uint4*>(&r);
#pragma unroll
        for (int k = 0; k < 8; ++k) z4[k] = make_uint4(0, 0, 0, 0);
    }
    if (in_range) {
        bool ok = !(v[3] <= 1e-12 || v[7] <= 1e-12 || v[11] <= 1e-12);                     // :94 (:97 is dead)
        double ndc[12];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int k = 0; k < 3; ++k) ndc[4 * q + k] = v[4 * q + k] / v[4 * q + 3];      // :101
            // w / w is exactly 1.0 for every finite non-zero w and NaN for an infinite one
            ndc[4 * q + 3] = __builtin_isfinite(v[4 * q + 3]) ? 1.0 : __builtin_nan("");
        }
        bool zo0 = (ndc[2] < -1.0 || ndc[2] > 1.0), zo1 = (ndc[6] < -1.0 || ndc[6] > 1.0),
             zo2 = (ndc[10] < -1.0 || ndc[10] > 1.0);
        ok = ok && !(zo0 && zo1 && zo2);                                                   // :103-106
#pragma unroll
        for (int k = 0; k < 12; ++k) ok = ok && __builtin_isfinite(ndc[k]);                // :109-114
        double sx[3], sy[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) { sx[q] = dot4(fp.vp, ndc + 4 * q); sy[q] = dot4(fp.vp + 4, ndc + 4 * q); }  // :117-121
        double e1x = sx[1] - sx[0], e1y = sy[1] - sy[0], e2x = sx[2] - sx[0], e2y = sy[2] - sy[0];
        double cross_product = e1x * e2y - e1y * e2x;                                      // :124-126
        ok = ok && !(cross_product <= 0);                                                  // :127
        int min_x_px = max(0, x86_cvttsd2si(floor(dmin3(sx[0], sx[1], sx[2]))));           // :130-133
        int max_x_px = min(fp.W - 1, x86_cvttsd2si(ceil(dmax3(sx[0], sx[1], sx[2]))));
        int min_y_px = max(0, x86_cvttsd2si(floor(dmin3(sy[0], sy[1], sy[2]))));
        int max_y_px = min(fp.H - 1, x86_cvttsd2si(ceil(dmax3(sy[0], sy[1], sy[2]))));
        ok = ok && !(min_x_px > max_x_px || min_y_px > max_y_px);                          // :135
        if (ok) {
            bx0 = min_x_px; by0 = min_y_px; bx1 = max_x_px; by1 = max_y_px;                // :138-141
            r.ax = sx[0]; r.ay = sy[0];
            r.s0x = sx[2] - sx[0]; r.s0y = sx[1] - sx[0];                                  // :78
            r.s1x = sy[2] - sy[0]; r.s1y = sy[1] - sy[0];                                  // :79
            r.uz = r.s0x * r.s1y - r.s0y * r.s1x;                                          // :80 (cross().z)
            // "Well scaled": every screen coordinate below 2^200 in magnitude, every edge delta zero or at
            // least 2^-250, u.z negative (it is -cross_product) and not tiny.  Then no product, sum or
            // quotient of the pixel loop can overflow, underflow or be NaN, which is what the division-free
            // coverage test and the FMA division by u.z in k_raster rely on ("exactness").
            {
                const double BIG = 0x1p200, SMALL = 0x1p-250;
                bool ws = r.uz < 0.0 && !(fabs(r.uz) < 1e-12);
#pragma unroll
                for (int q = 0; q < 3; ++q) ws = ws && fabs(sx[q]) < BIG && fabs(sy[q]) < BIG;
                const double dl[4] = { r.s0x, r.s0y, r.s1x, r.s1y };
#pragma unroll
                for (int q = 0; q < 4; ++q) ws = ws && (dl[q] == 0.0 || fabs(dl[q]) >= SMALL);
                // ... and not a sliver whose coverage roundings could reach past its bounding box.  k_raster's fast visit does not test
                // a pixel against the bbox of :130-133: with every computed u within E = 2^-50 S R of its exact value (S = sum of the
                // |edge deltas|, R >= |A - pixel centre| (L1) on the blocks the bbox touches) and u.z within 2^-52 S^2, a pixel that
                // passes the sign test has exact barycentrics >= -eps, eps < 2^-48 S R / |u.z|, so it lies within 2 eps S of the
                // triangle's extent in x and in y - while a pixel centre outside floor(min) .. ceil(max) is at least 0.5 away from it.
                // 2^-40 S^2 R < |u.z| (R >= S) makes 2 eps S < 2^-7.  The others take the literal path, which tests the bbox.
                {
                    const double S = (fabs(r.s0x) + fabs(r.s0y)) + (fabs(r.s1x) + fabs(r.s1y));
                    const double rx = fmax(fabs(r.ax - ((double)bx0 + 0.5)), fabs(r.ax - ((double)bx1 + 0.5)));
                    const double ry = fmax(fabs(r.ay - ((double)by0 + 0.5)), fabs(r.ay - ((double)by1 + 0.5)));
                    ws = ws && 0x1p-40 * (S * S) * (rx + ry + 17.0 + S) < fabs(r.uz);
                }
                r.ruz = ws ? 1.0 / r.uz : 0.0;
            }
            r.z0 = ndc[2]; r.z1 = ndc[6]; r.z2 = ndc[10];
            r.bx0 = (uint16_t)bx0; r.by0 = (uint16_t)by0; r.bx1 = (uint16_t)bx1; r.by1 = (uint16_t)by1;
            r.color = d.colors ? d.colors[i] : 0xffffffffu;
            r.dl = ((uint32_t)draw_idx << 24) | i | (r.ruz == 0.0 ? TRGL_DL_LITERAL : 0u);
            // Depth plane of the early depth test (k_raster).  The z of our_gl.cpp:156-158 is the plane
            //   z0 + ((ax - x) Gx + (ay - y) Gy) / u.z,  Gx = s1x (z1-z0) - s1y (z2-z0),  Gy = s0y (z2-z0) - s0x (z1-z0)
            // through the three vertices up to the roundings of u.x, u.y, the three quotients and the weighted sum: at most
            // 2^-50 max|z_i| (R S / |u.z| + 1) for a covered pixel, where R >= |A - pixel centre| (L1) over the clamped bbox (and R >= S)
            // and S = the sum of the |edge deltas|.  g1 = Gx/u.z and g2 = Gy/u.z carry a few more roundings of that size, so does the
            // evaluation c0 + (ax - x) g1 + (ay - y) g2 itself, and c0 = z0 - 2^-39 max|z_i| (R S / |u.z| + 1) covers all of it a
            // thousand times over: a pixel whose plane value is >= the stored depth fails the strict `<` of :165 whatever its
            // coverage and low bits.  Switched off (c0 = -inf, g = 0) for triangles that are not well scaled and when a constant
            // could leave the normal range.
            r.c0 = -__builtin_inf(); r.g1 = 0.0; r.g2 = 0.0;
            if (r.ruz != 0.0) {
                const double zabs = dmax3(fabs(r.z0), fabs(r.z1), fabs(r.z2));
                const double dz1 = r.z1 - r.z0, dz2 = r.z2 - r.z0;
                const double h1 = (r.s1x * dz1 - r.s1y * dz2) * r.ruz, h2 = (r.s0y * dz2 - r.s0x * dz1) * r.ruz;
                const double rx = fmax(fabs(r.ax - ((double)bx0 + 0.5)), fabs(r.ax - ((double)bx1 + 0.5)));
                const double ry = fmax(fabs(r.ay - ((double)by0 + 0.5)), fabs(r.ay - ((double)by1 + 0.5)));
                const double Ssum = (fabs(r.s0x) + fabs(r.s0y)) + (fabs(r.s1x) + fabs(r.s1y));
                const double R = rx + ry + 1.0 + Ssum;       // (+ S: the plane is also evaluated at the three vertices, k_raster's cull)
                const double mz = zabs * 0x1p-39 * (R * Ssum * fabs(r.ruz) + 1.0) + 0x1p-600;
                // trusted only while nothing can overflow (R |g| bounds each product of the test); NaN compares false
                const bool okp = zabs < 0x1p1000 && R * fabs(h1) < 0x1p900 && R * fabs(h2) < 0x1p900 && mz < 0x1p1000;
                if (okp) { r.c0 = r.z0 - mz; r.g1 = h1; r.g2 = h2; }
            }
            if (d.kind != TRGL_SHADER_FLAT) {                                                  // :168-170
                TriW w;
                w.iw0 = (fabs(v[3]) > 1e-12) ? (1.0 / v[3]) : 0.0;
                w.iw1 = (fabs(v[7]) > 1e-12) ? (1.0 / v[7]) : 0.0;
                w.iw2 = (fabs(v[11]) > 1e-12) ? (1.0 / v[11]) : 0.0;
                w.pad = 0.0;
                recs_w[d.first + i] = w;
            }
            // barycentric() rejects every pixel when |u.z| < 1e-12 (our_gl.cpp:82-83): no pairs then.
            // Rows outside this context's strip are not ours either.
            int y_lo = max(by0, fp.strip_y0), y_hi = min(by1, fp.strip_y1 - 1);
            if (!(fabs(r.uz) < 1e-12) && y_lo <= y_hi) {
                uint32_t tx0 = bx0 >> TRGL_TILE_LOG2, tx1 = bx1 >> TRGL_TILE_LOG2;
                uint32_t ty0 = y_lo >> TRGL_TILE_LOG2, ty1 = y_hi >> TRGL_TILE_LOG2;
                // tile rows of the box that this context owns: all of them for a strip (the box is clipped to it), the ones of
                // its bands with interleaved ownership (k_expand walks the same rows)
                const uint32_t rows = fp.il_tiles ? (uint32_t)(il_owned_below(fp, (int)ty1 + 1) - il_owned_below(fp, (int)ty0)) : ty1 - ty0 + 1;
                ntiles = (tx1 - tx0 + 1) * rows;
                // the box in BLOCK units (8 x 8 pixels; tile = block >> 2): k_expand derives from it, for every tile of the box,
                // the 4 x 4 mask of the tile's blocks that the box reaches
                tb = make_uint2(((uint32_t)bx0 >> TRGL_BLOCK_LOG2) | (((uint32_t)y_lo >> TRGL_BLOCK_LOG2) << 16),
                                ((uint32_t)bx1 >> TRGL_BLOCK_LOG2) | (((uint32_t)y_hi >> TRGL_BLOCK_LOG2) << 16));
                // A 7-bit lower bound of the triangle's depths rides with every pair (in the spare bits of its triangle word, k_expand):
                // every covered pixel has z above the depth plane's smallest value at the three vertices (k_raster's cull explains
                // why), zv, hence z >= -1 + zq / 64 for zq = floor((zv + 1) 64 - 2^-20).  In the spare bits of the box: block
                // coordinates stay below 2^13.  A flush that holds large triangles (`large`) makes k_raster's list steps use it: an
                // entry whose bound is not below the block's largest depth never becomes a candidate - no gather, no test.
                uint32_t zq = 0;
                if (r.c0 > -__builtin_inf()) {
                    const double zv = dmin3(r.c0, __builtin_fma(-r.s0y, r.g1, __builtin_fma(-r.s1y, r.g2, r.c0)),
                                            __builtin_fma(-r.s0x, r.g1, __builtin_fma(-r.s1x, r.g2, r.c0)));
                    const double t = (zv + 1.0) * 64.0 - 0x1p-20;
                    if (t >= 1.0) zq = t >= 127.0 ? 127u : (uint32_t)x86_cvttsd2si(floor(t));
                }
                tb.x |= (zq & 7u) << 13; tb.y |= ((zq >> 3) & 7u) << 13; tb.y |= (zq >> 6) << 29;
                large = bx1 - bx0 >= 64 || y_hi - y_lo >= 64;
            }
        }
        cnt[d.first + i] = ntiles;
        tilebox[d.first + i] = tb;
    }
    // ---- record stream out: each wave stages its own 64 records (8 KB) in LDS, 16-B chunks XOR-swizzled so neither the
    // per-thread writes (128-B stride) nor the linear read-out conflict on banks, and writes them as one contiguous run.
    // A record is only ever read through a (tile, triangle) pair: triangles without pairs (rejected, or outside this
    // context's strip - 7 of 8 on an 8-GPU shard) need no 128-B store; which ones is a wave ballot, so the phase needs no
    // block barrier and no LDS beyond the 32 KB staging buffer (5 blocks per CU instead of 4).
    const uint32_t lane = tid & 63, wbase = tid & ~63u;
    {
        uint4* l4 = reinterpret_cast<uint4*>(s_buf);
        const uint4* r4 = reinterpret_cast<const uint4*>(&r);
#pragma unroll
        for (int c = 0; c < 8; ++c) l4[tid * 8 + (c ^ (tid & 7))] = r4[c];
        const unsigned long long keep = __ballot(ntiles != 0);
        __builtin_amdgcn_wave_barrier();
        uint4* dst = reinterpret_cast<uint4*>(recs + d.first + b0 + wbase);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const uint32_t k = lane + 64u * it, t = k >> 3, c = k & 7;            // chunk c of the wave's record t
            if ((keep >> t) & 1ull) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(&l4[(wbase + t) * 8 + (c ^ (t & 7))]), reinterpret_cast<u32x4*>(&dst[k]));
            }
        }
    }
    // pairs of this block of 256 triangles: k_expand derives every triangle's slice of the pair list from these sums
    // (k_chunk_spine) and a block-level scan of `cnt`, so no scan pass ever walks the N-element arrays
    {
        uint32_t ws = ntiles;
        for (int o = 32; o; o >>= 1) ws += __shfl_xor(ws, o);
        __syncthreads();                                   // every wave is done with the staging buffer
        uint32_t* s_ws = reinterpret_cast<uint32_t*>(s_buf);
        if (lane == 0) s_ws[tid >> 6] = ws;
        __syncthreads();
        if (tid == 0) blk_sums[blk_base + blockIdx.x] = s_ws[0] + s_ws[1] + s_ws[2] + s_ws[3];
    }
    // triangles with pairs that take the literal (dividing) path of k_raster: none in any realistic frame, and then the host
    // launches the raster kernel that does not contain that path (trgl_flush_end)
    {
        const unsigned long long lit = __ballot(ntiles != 0 && r.ruz == 0.0);
        if (lit && lane == 0) atomicAdd(&stats->literal_tris, (unsigned long long)__popcll(lit));
        const unsigned long long lrg = __ballot(large);
        if (lrg && lane == 0) atomicAdd(&stats->large_tris, (unsigned long long)__popcll(lrg));
    }
    // bbox stats (our_gl.cpp:138-141): one set of atomics per wave
    int wx0 = wave_min_i(bx0), wy0 = wave_min_i(by0), wx1 = wave_max_i(bx1), wy1 = wave_max_i(by1);
    // All waves hit the same four words, and same-address atomics serialise at ~11 ns each, so only
    // issue one when it would change the value.  The plain loads may be stale, but min only falls and
    // max only rises: a stale value can cause a redundant atomic, never a missed one.
    if ((threadIdx.x & 63) == 0 && wx0 != INT_MAX) {
        if (wx0 < __builtin_nontemporal_load(&stats->min_x)) atomicMin(&stats->min_x, wx0);
        if (wy0 < __builtin_nontemporal_load(&stats->min_y)) atomicMin(&stats->min_y, wy0);
        if (wx1 > __builtin_nontemporal_load(&stats->max_x)) atomicMax(&stats->max_x, wx1);
        if (wy1 > __builtin_nontemporal_load(&stats->max_y)) atomicMax(&stats->max_y, wy1);
    }
}

// ---------------------------------------------------------------------------------------------
// block-level scan helpers (the radix histograms are scanned row by row in k_radix_scan_rows)
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(v, o); if (lane >= o) v += t; }
    return v;
}

// block-wide exclusive scan of one value per thread (256 threads); returns exclusive prefix, total in *total
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* smem /*[4]*/, uint32_t* total) {
    uint32_t inc = wave_incl_scan(v);
    int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 63) smem[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < SCAN_THREADS / 64; ++k) { uint32_t s = smem[k]; if (k < w) base += s; tot += s; }
    *total = tot;
    return base + inc - v;
}

// ---------------------------------------------------------------------------------------------
// pair offsets.  k_setup leaves one pair count per block of 256 triangles (blk_sums, blocks numbered across the
// draws of the flush).  k_chunk_spine (one block) scans them in chunks of 16: chunk_off[c] = pairs before block 16c.
// k_expand then finds its block's base (chunk offset + the <= 15 preceding block sums of its chunk) and scans the 256
// counts of its own triangles in LDS.
// ---------------------------------------------------------------------------------------------
constexpr int EXPAND_CHUNK = 16;

constexpr int SPINE_THREADS = 1024;
__global__ __launch_bounds__(SPINE_THREADS) void k_chunk_spine(const uint32_t* __restrict__ blk_sums, uint32_t nblk,
                                                               uint32_t* __restrict__ chunk_off,
                                                               unsigned long long* __restrict__ total64, unsigned long long* __restrict__ host_copy,
                                                               uint4* __restrict__ zero16, uint32_t n_zero16) {
    __shared__ unsigned long long s_wave[SPINE_THREADS / 64];
    // (the tile bounds of the flush start from zero: cleared here instead of by a fill command of its own on the stream, 4.6 us)
    for (uint32_t k = threadIdx.x; k < n_zero16; k += SPINE_THREADS) zero16[k] = make_uint4(0, 0, 0, 0);
    const uint32_t nchunks = (nblk + EXPAND_CHUNK - 1) / EXPAND_CHUNK;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // One chunk (16 block sums = 64 bytes) per thread, 1024 chunks per round.  Offsets are 32-bit by design (a flush holds
    // fewer than 2^32 pairs); everything is accumulated in 64 bits so that a flush beyond that limit is seen as such by the
    // host and by the guards of the kernels queued behind this one instead of wrapping (100 k full-screen triangles at
    // 8192^2 are 6.5e9 pairs).
    unsigned long long running = 0;
    for (uint32_t start = 0; start < nchunks; start += SPINE_THREADS) {
        const uint32_t c = start + threadIdx.x;
        unsigned long long v = 0;
        if (c < nchunks) {
            const uint32_t q0 = c * EXPAND_CHUNK;
            if (q0 + EXPAND_CHUNK <= nblk) {
                const uint4* p4 = reinterpret_cast<const uint4*>(blk_sums + q0);
#pragma unroll
                for (int k = 0; k < EXPAND_CHUNK / 4; ++k) { const uint4 t = p4[k]; v += (unsigned long long)t.x + t.y + t.z + t.w; }
            } else {
                for (uint32_t q = q0; q < nblk; ++q) v += blk_sums[q];
            }
        }
        unsigned long long inc = v;                                   // inclusive scan inside the wave
        for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane == 63) s_wave[w] = inc;
        __syncthreads();
        unsigned long long base = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < SPINE_THREADS / 64; ++k) { const unsigned long long t = s_wave[k]; if (k < w) base += t; tot += t; }
        if (c < nchunks) chunk_off[c] = (uint32_t)(running + base + inc - v);
        running += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *total64 = running;
        // The host needs the pair count and the two triangle counts that k_setup left next to it (DevStats: literal_tris, large_tris)
        // before it launches the raster: written straight into its pinned memory instead of a copy command on the stream (4.6 us).
        host_copy[0] = running; host_copy[1] = total64[1]; host_copy[2] = total64[2];
        __threadfence_system();
    }
}

// expand: triangle i owns pairs [off, off + cnt[i]) = its tiles in row-major order; off is computed here.
// Triangles with many tiles are written by the whole wave.  One launch per draw (same blocks as k_setup).
// The pairs of a block are consecutive in the output (the offsets are a prefix sum in submission order), so they are
// assembled in LDS and written out linearly: whole cache lines instead of 64 lanes x a few elements each with a stride.
// A block whose triangles cover more than EXPAND_STAGE tiles writes straight to memory.
// Every pair also carries the 4 x 4 mask of the tile's 8 x 8-pixel blocks that the triangle's clamped bbox reaches (bit 4 cy + cx):
// the raster wave that owns a block picks its candidates from the tile's list by that bit, without touching their records.
// K: the key type, 16 bits while the frame has at most 65536 tiles (up to 8192x8192), 32 bits beyond.
constexpr uint32_t EXPAND_STAGE = 3072;
// blocks of tile (tx, ty) inside the block-unit box [qx0, qx1] x [qy0, qy1] (the box reaches the tile)
__device__ __forceinline__ uint32_t tile_block_mask(uint32_t tx, uint32_t ty, uint32_t qx0, uint32_t qy0, uint32_t qx1, uint32_t qy1) {
    const uint32_t c0 = qx0 > 4 * tx ? qx0 - 4 * tx : 0u, c1 = qx1 < 4 * tx + 3 ? qx1 - 4 * tx : 3u;
    const uint32_t r0 = qy0 > 4 * ty ? qy0 - 4 * ty : 0u, r1 = qy1 < 4 * ty + 3 ? qy1 - 4 * ty : 3u;
    return (((2u << c1) - (1u << c0)) & 0xfu) * 0x1111u & ((0xffffu >> (12 - 4 * r1)) & (0xffffu << (4 * r0)));
}
template <typename K>
__global__ __launch_bounds__(256) void k_expand(FrameParams fp, uint32_t first, uint32_t n, int tiles_x, const uint32_t* __restrict__ cnt,
                                                const uint32_t* __restrict__ blk_sums, const uint32_t* __restrict__ chunk_off,
                                                uint32_t blk_base, const uint2* __restrict__ tilebox,
                                                K* __restrict__ keys, uint32_t* __restrict__ vals, uint16_t* __restrict__ bmask,
                                                const unsigned long long* __restrict__ pairs_total, uint32_t cap) {
    __shared__ uint32_t smem[4];
    __shared__ uint32_t s_v[EXPAND_STAGE];
    __shared__ K s_k[EXPAND_STAGE];
    __shared__ uint16_t s_m[EXPAND_STAGE];
    if (*pairs_total > cap) return;        // the host sized the buffers from an earlier flush: it will grow them and launch again
    const uint32_t local = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = first + local;                          // index of the triangle within the flush
    const uint32_t blk = blk_base + blockIdx.x;
    uint32_t c = 0; uint2 tb = make_uint2(0, 0);
    if (local < n) { c = cnt[i]; if (c) tb = tilebox[i]; }
    // pairs before this block = those before its chunk of 16 blocks (k_chunk_spine) + the sums of the blocks before it inside the chunk:
    // ONE load by up to 15 lanes and four butterfly steps (a loop of up to 15 dependent wave-uniform loads cost ~1 us each)
    uint32_t base;
    {
        const uint32_t q0 = blk - blk % EXPAND_CHUNK, l = threadIdx.x & 63u;
        uint32_t part = l < blk - q0 ? blk_sums[q0 + l] : 0u;
        for (int o = 8; o; o >>= 1) part += __shfl_xor(part, o);
        base = chunk_off[blk / EXPAND_CHUNK] + (uint32_t)__builtin_amdgcn_readfirstlane((int)part);
    }
    uint32_t tot;
    const uint32_t o = block_excl_scan(c, smem, &tot);         // offset inside the block's run of pairs
    const bool staged = tot <= EXPAND_STAGE;                   // block-uniform
    K* const kdst = keys + base; uint32_t* const vdst = vals + base; uint16_t* const mdst = bmask + base;
    constexpr uint32_t SMALL = 8;
    if (c && c <= SMALL) {
        // row-major walk with running counters instead of a division and a modulo per pair
        const uint32_t qx0 = tb.x & 0x1fff, qy0 = (tb.x >> 16) & 0x1fff, qx1 = tb.y & 0x1fff, qy1 = (tb.y >> 16) & 0x1fff;
        const uint32_t iz = i | ((((tb.x >> 13) & 7u) | (((tb.y >> 13) & 7u) << 3) | ((tb.y >> 29) << 6)) << 25);      // TRGL_VAL_TRI | TRGL_VAL_ZQ
        const uint32_t tx0 = qx0 >> 2, ty0 = qy0 >> 2, tx1 = qx1 >> 2;
        uint32_t tx = tx0, row = 0;
        uint32_t ty = fp.il_tiles ? (uint32_t)il_nth_owned_from(fp, (int)ty0, 0) : ty0;
        for (uint32_t k = 0; k < c; ++k) {
            const K key = (K)(ty * tiles_x + tx);
            const uint16_t m = (uint16_t)tile_block_mask(tx, ty, qx0, qy0, qx1, qy1);
            if (staged) { s_k[o + k] = key; s_v[o + k] = iz; s_m[o + k] = m; } else { kdst[o + k] = key; vdst[o + k] = iz; mdst[o + k] = m; }
            if (++tx > tx1) { tx = tx0; ++row; ty = fp.il_tiles ? (uint32_t)il_nth_owned_from(fp, (int)ty0, (int)row) : ty0 + row; }
        }
    }
    unsigned long long big = __ballot(c > SMALL);
    int lane = threadIdx.x & 63;
    while (big) {
        int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        uint32_t cc = __shfl(c, src), oo = __shfl(o, src), ii = __shfl(i, src);
        uint32_t bx = __shfl(tb.x, src), by = __shfl(tb.y, src);
        const uint32_t qx0 = bx & 0x1fff, qy0 = (bx >> 16) & 0x1fff, qx1 = by & 0x1fff, qy1 = (by >> 16) & 0x1fff;
        ii |= (((bx >> 13) & 7u) | (((by >> 13) & 7u) << 3) | ((by >> 29) << 6)) << 25;
        uint32_t tx0 = qx0 >> 2, ty0 = qy0 >> 2, tx1 = qx1 >> 2;
        uint32_t wdt = tx1 - tx0 + 1;
        for (uint32_t k = lane; k < cc; k += 64) {
            uint32_t ty = fp.il_tiles ? (uint32_t)il_nth_owned_from(fp, (int)ty0, (int)(k / wdt)) : ty0 + k / wdt, tx = tx0 + k % wdt;
            const K key = (K)(ty * tiles_x + tx);
            const uint16_t m = (uint16_t)tile_block_mask(tx, ty, qx0, qy0, qx1, qy1);
            if (staged) { s_k[oo + k] = key; s_v[oo + k] = ii; s_m[oo + k] = m; } else { kdst[oo + k] = key; vdst[oo + k] = ii; mdst[oo + k] = m; }
        }
    }
    if (staged) {
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < tot; j += 256) { kdst[j] = s_k[j]; vdst[j] = s_v[j]; mdst[j] = s_m[j]; }
    }
}

// ---------------------------------------------------------------------------------------------
// stable LSD radix pass on `bits` key bits at `shift`.  A 256-thread block owns RADIX_CHUNK consecutive
// pairs; wave w owns the w-th quarter and walks it in order, 64 pairs a step, ranking equal digits
// with ballots, so equal keys keep their input order (= triangle submission order).  The chunk is
// then reordered by digit in LDS and written out linearly, so global stores are contiguous runs
// (RADIX_CHUNK / 2^bits pairs per digit on average) instead of 64 scattered 4-byte elements per
// instruction.  hist layout: [digit][block] so one exclusive scan yields every block's base per digit.
// ---------------------------------------------------------------------------------------------
constexpr int RADIX_CHUNK = 4096;
constexpr int RADIX_WAVE_CHUNK = RADIX_CHUNK / 4;
constexpr int RADIX_ROUNDS = RADIX_WAVE_CHUNK / 64;      // 16
constexpr int RADIX_MAX_BITS = 8;

// four consecutive keys with one load (the pair buffers hold a multiple of 4 entries and p is a multiple of 4)
template <typename K> __device__ __forceinline__ void load4(const K* __restrict__ keys, uint64_t p, uint32_t k[4]);
template <> __device__ __forceinline__ void load4<uint32_t>(const uint32_t* __restrict__ keys, uint64_t p, uint32_t k[4]) {
    const uint4 q = *reinterpret_cast<const uint4*>(keys + p);
    k[0] = q.x; k[1] = q.y; k[2] = q.z; k[3] = q.w;
}
template <> __device__ __forceinline__ void load4<uint16_t>(const uint16_t* __restrict__ keys, uint64_t p, uint32_t k[4]) {
    const uint2 q = *reinterpret_cast<const uint2*>(keys + p);
    k[0] = q.x & 0xffffu; k[1] = q.x >> 16; k[2] = q.y & 0xffffu; k[3] = q.y >> 16;
}

template <typename K>
__global__ __launch_bounds__(256) void k_radix_hist(const K* __restrict__ keys, const unsigned long long* __restrict__ pairs_total,
                                                    uint32_t cap, int shift, int bits,
                                                    uint32_t nblocks, uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_cnt[1 << RADIX_MAX_BITS];
    const unsigned long long P64 = *pairs_total;
    const uint32_t P = P64 > cap ? 0u : (uint32_t)P64;      // over capacity: nothing was expanded; every block counts nothing
    const uint32_t nb = 1u << bits, mask = nb - 1;
    for (uint32_t b = threadIdx.x; b < nb; b += 256) s_cnt[b] = 0;
    __syncthreads();
    const uint64_t beg = (uint64_t)blockIdx.x * RADIX_CHUNK;
    uint64_t end = beg + RADIX_CHUNK; if (end > P) end = P;      // blocks past the last pair (the grid covers the capacity) add zeros
    for (uint64_t p = beg + 4u * threadIdx.x; p < end; p += 1024) {
        uint32_t k[4];
        load4<K>(keys, p, k);
#pragma unroll
        for (int q = 0; q < 4; ++q) if (p + q < end) atomicAdd(&s_cnt[(k[q] >> shift) & mask], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += 256) hist[(size_t)b * nblocks + blockIdx.x] = s_cnt[b];
}

// One block per digit: exclusive scan of the digit's row of per-block counts, in place ([digit][block] layout: a row is contiguous),
// and the row's total to totals[digit].  k_radix_scatter adds the exclusive scan of the (at most 256) totals itself.  One launch instead
// of the three of a generic scan over the whole table (17 -> 6 us per pass on the C4 frame).
constexpr int ROWSCAN_THREADS = 1024, ROWSCAN_PER = 8;
__global__ __launch_bounds__(ROWSCAN_THREADS) void k_radix_scan_rows(uint32_t* __restrict__ hist, uint32_t nblocks, uint32_t* __restrict__ totals) {
    __shared__ uint32_t s_wave[ROWSCAN_THREADS / 64];
    uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t running = 0;
    for (uint32_t start = 0; start < nblocks; start += ROWSCAN_THREADS * ROWSCAN_PER) {
        const uint32_t base = start + threadIdx.x * ROWSCAN_PER;
        uint32_t v[ROWSCAN_PER], sum = 0;
#pragma unroll
        for (int k = 0; k < ROWSCAN_PER; ++k) { v[k] = (base + k < nblocks) ? row[base + k] : 0; sum += v[k]; }
        uint32_t inc = sum;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane == 63) s_wave[w] = inc;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < ROWSCAN_THREADS / 64; ++k) { const uint32_t t = s_wave[k]; if (k < w) wbase += t; tot += t; }
        uint32_t run = running + wbase + inc - sum;
#pragma unroll
        for (int k = 0; k < ROWSCAN_PER; ++k) { if (base + k < nblocks) row[base + k] = run; run += v[k]; }
        running += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = running;
}

template <typename K>
__global__ __launch_bounds__(256) void k_radix_scatter(const K* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                       const uint16_t* __restrict__ msk_in,
                                                       const unsigned long long* __restrict__ pairs_total, uint32_t cap,
                                                       int shift, int bits, uint32_t nblocks,
                                                       const uint32_t* __restrict__ base, const uint32_t* __restrict__ totals,
                                                       K* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint16_t* __restrict__ msk_out) {
    __shared__ uint32_t s_cnt[4][1 << RADIX_MAX_BITS];     // per wave: running count, then (after phase 2) local start
    __shared__ uint32_t s_start[1 << RADIX_MAX_BITS];      // first local position of each digit in the chunk
    __shared__ uint32_t s_gbase[1 << RADIX_MAX_BITS];      // global position of the chunk's first pair of each digit
    __shared__ uint32_t s_val[RADIX_CHUNK];
    __shared__ K s_key[RADIX_CHUNK];
    __shared__ uint16_t s_msk[RADIX_CHUNK];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long P64 = *pairs_total;
    const uint32_t P = P64 > cap ? 0u : (uint32_t)P64;
    if ((uint64_t)blockIdx.x * RADIX_CHUNK >= P) return;     // the grid covers the capacity of the buffers, not the pairs of this flush
    const uint32_t nb = 1u << bits, mask = nb - 1;
    for (uint32_t b = threadIdx.x; b < nb; b += 256) {
        s_cnt[0][b] = 0; s_cnt[1][b] = 0; s_cnt[2][b] = 0; s_cnt[3][b] = 0;
        s_gbase[b] = base[(size_t)b * nblocks + blockIdx.x];      // pairs of this digit in the blocks before this one (k_radix_scan_rows)
    }
    if (w == 0) {       // + the pairs of all smaller digits: exclusive scan of the row totals (nb <= 256: up to 4 digits per lane)
        uint32_t tot[4], run = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const uint32_t d = lane * 4 + q; tot[q] = d < nb ? totals[d] : 0; run += tot[q]; }
        uint32_t inc = run;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        uint32_t ex = inc - run;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const uint32_t d = lane * 4 + q; if (d < nb) s_start[d] = ex; ex += tot[q]; }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += 256) s_gbase[b] += s_start[b];
    __syncthreads();
    const uint64_t cbeg = (uint64_t)blockIdx.x * RADIX_CHUNK;
    uint64_t cend = cbeg + RADIX_CHUNK; if (cend > P) cend = P;
    const uint32_t n_chunk = (uint32_t)(cend - cbeg);
    const uint64_t wbeg = cbeg + (uint64_t)w * RADIX_WAVE_CHUNK;
    const unsigned long long lt = (1ull << lane) - 1ull;

    // ---- phase 1: stable rank of every pair among the equal digits of its wave's quarter ------------
    // (key and block mask share a register: key in the low 16 bits when K is 16 bits wide; else the mask rides in its own)
    uint32_t k[RADIX_ROUNDS], v[RADIX_ROUNDS], rk[RADIX_ROUNDS];
    uint16_t mk[RADIX_ROUNDS];
#pragma unroll
    for (int r = 0; r < RADIX_ROUNDS; ++r) {
        const uint64_t p = wbeg + (uint64_t)r * 64 + lane;
        const bool act = p < cend;
        k[r] = act ? (uint32_t)keys_in[p] : 0; v[r] = act ? vals_in[p] : 0; mk[r] = act ? msk_in[p] : (uint16_t)0;
    }
#pragma unroll
    for (int r = 0; r < RADIX_ROUNDS; ++r) {
        const bool act = wbeg + (uint64_t)r * 64 + lane < cend;
        const uint32_t dgt = (k[r] >> shift) & mask;
        unsigned long long same = __ballot(act);
        for (int b = 0; b < bits; ++b) {
            const unsigned long long bal = __ballot((dgt >> b) & 1u);
            same &= ((dgt >> b) & 1u) ? bal : ~bal;
        }
        const uint32_t cur = act ? s_cnt[w][dgt] : 0;
        __builtin_amdgcn_wave_barrier();
        const uint32_t r_in = __popcll(same & lt);
        rk[r] = cur + r_in;
        if (act && r_in == 0) s_cnt[w][dgt] = cur + (uint32_t)__popcll(same);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- phase 2: local layout: digits ascending, inside a digit waves ascending ------------------------
    if (w == 0) {
        // exclusive scan over digits of the chunk totals (nb <= 256: up to 4 digits per lane)
        uint32_t tot[4], run = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t d = lane * 4 + q;
            tot[q] = d < nb ? s_cnt[0][d] + s_cnt[1][d] + s_cnt[2][d] + s_cnt[3][d] : 0;
            run += tot[q];
        }
        uint32_t inc = run;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        uint32_t ex = inc - run;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const uint32_t d = lane * 4 + q; if (d < nb) s_start[d] = ex; ex += tot[q]; }
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < nb; d += 256) {
        uint32_t run = s_start[d];
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) { const uint32_t c = s_cnt[ww][d]; s_cnt[ww][d] = run; run += c; }
    }
    __syncthreads();
    // ---- phase 3: reorder in LDS ------------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < RADIX_ROUNDS; ++r) {
        if (wbeg + (uint64_t)r * 64 + lane < cend) {
            const uint32_t dgt = (k[r] >> shift) & mask;
            const uint32_t lp = s_cnt[w][dgt] + rk[r];
            s_key[lp] = (K)k[r]; s_val[lp] = v[r]; s_msk[lp] = mk[r];
        }
    }
    __syncthreads();
    // ---- phase 4: linear read-out, contiguous global runs per digit ------------------------------------------
    for (uint32_t i = threadIdx.x; i < n_chunk; i += 256) {
        const uint32_t key = s_key[i];
        const uint32_t dgt = (key >> shift) & mask;
        const uint32_t dst = s_gbase[dgt] + (i - s_start[dgt]);
        keys_out[dst] = (K)key; vals_out[dst] = s_val[i]; msk_out[dst] = s_msk[i];
    }
}

// per-tile slice [start, end) of the sorted pair list: four consecutive pairs per thread (one load)
template <typename K>
__global__ __launch_bounds__(256) void k_bounds(const K* __restrict__ keys, const unsigned long long* __restrict__ pairs_total, uint32_t cap,
                                                uint32_t* __restrict__ tile_start, uint32_t* __restrict__ tile_end) {
    const unsigned long long P64 = *pairs_total;
    const uint32_t P = P64 > cap ? 0u : (uint32_t)P64;
    const uint32_t p0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
    if (p0 >= P) return;
    uint32_t k[4];
    load4<K>(keys, p0, k);                                              // the buffers hold a multiple of 4 entries (grow_pairs)
    uint32_t prev = p0 ? (uint32_t)keys[p0 - 1] : 0u;
    const uint32_t after = (p0 + 4 < P) ? (uint32_t)keys[p0 + 4] : 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t p = p0 + (uint32_t)i;
        if (p < P) {
            const uint32_t next = i < 3 ? k[i + 1] : after;
            if (p == 0 || prev != k[i]) tile_start[k[i]] = p;
            if (p == P - 1 || next != k[i]) tile_end[k[i]] = p + 1;
        }
        prev = k[i];
    }
}

}  // namespace

// ---- launchers ----------------------------------------------------------------------------------
namespace trgl {

uint32_t setup_num_blocks(uint32_t n) { return (n + SETUP_THREADS - 1) / SETUP_THREADS; }

void launch_setup(hipStream_t s, const FrameParams& fp, const DrawDesc& draw, DrawDesc* draws_dev, int draw_idx, uint32_t n,
                  TriRec* recs, TriW* recs_w, uint32_t* cnt, uint2* tilebox, DevStats* stats, uint32_t* blk_sums, uint32_t blk_base) {
    if (!n) return;
    hipLaunchKernelGGL(k_setup, dim3(setup_num_blocks(n)), dim3(SETUP_THREADS), 0, s, fp, draw, draws_dev, draw_idx, recs, recs_w, cnt, tilebox,
                       stats, blk_sums, blk_base);
}

void launch_chunk_spine(hipStream_t s, const uint32_t* blk_sums, uint32_t nblk, uint32_t* chunk_off, unsigned long long* total64, unsigned long long* host_copy,
                        void* zero, size_t zero_bytes) {
    hipLaunchKernelGGL(k_chunk_spine, dim3(1), dim3(SPINE_THREADS), 0, s, blk_sums, nblk, chunk_off, total64, host_copy, (uint4*)zero, (uint32_t)(zero_bytes / 16));
}

void launch_expand(hipStream_t s, const FrameParams& fp, uint32_t first, uint32_t n, int tiles_x, const uint32_t* cnt, const uint32_t* blk_sums,
                   const uint32_t* chunk_off, uint32_t blk_base, const uint2* tilebox, void* keys, bool key16, uint32_t* vals, uint16_t* bmask,
                   const unsigned long long* pairs_total, uint32_t cap) {
    if (!n) return;
    if (key16)
        hipLaunchKernelGGL(k_expand<uint16_t>, dim3(setup_num_blocks(n)), dim3(SETUP_THREADS), 0, s, fp, first, n, tiles_x, cnt, blk_sums, chunk_off,
                           blk_base, tilebox, (uint16_t*)keys, vals, bmask, pairs_total, cap);
    else
        hipLaunchKernelGGL(k_expand<uint32_t>, dim3(setup_num_blocks(n)), dim3(SETUP_THREADS), 0, s, fp, first, n, tiles_x, cnt, blk_sums, chunk_off,
                           blk_base, tilebox, (uint32_t*)keys, vals, bmask, pairs_total, cap);
}

uint32_t radix_num_workers(uint32_t P) { return (uint32_t)(((uint64_t)P + RADIX_CHUNK - 1) / RADIX_CHUNK); }   // = blocks of a pass

// The pair count of the flush stays on the device (`pairs_total`): grids cover `cap`, the capacity of the pair buffers,
// and blocks past the last pair do nothing, so the host never has to wait for the count before it can queue these.
template <typename K>
static void radix_pass_t(hipStream_t s, const K* keys_in, const uint32_t* vals_in, const uint16_t* msk_in, K* keys_out, uint32_t* vals_out, uint16_t* msk_out,
                         const unsigned long long* pairs_total, uint32_t cap, int shift, int bits, uint32_t* hist, uint32_t* scan_tmp) {
    uint32_t nblk = radix_num_workers(cap);
    hipLaunchKernelGGL(k_radix_hist<K>, dim3(nblk), dim3(256), 0, s, keys_in, pairs_total, cap, shift, bits, nblk, hist);
    hipLaunchKernelGGL(k_radix_scan_rows, dim3(1u << bits), dim3(ROWSCAN_THREADS), 0, s, hist, nblk, scan_tmp);
    hipLaunchKernelGGL(k_radix_scatter<K>, dim3(nblk), dim3(256), 0, s, keys_in, vals_in, msk_in, pairs_total, cap, shift, bits, nblk, hist, scan_tmp,
                       keys_out, vals_out, msk_out);
}

void launch_radix_pass(hipStream_t s, const void* keys_in, const uint32_t* vals_in, const uint16_t* msk_in, void* keys_out,
                       uint32_t* vals_out, uint16_t* msk_out, bool key16, const unsigned long long* pairs_total, uint32_t cap, int shift, int bits,
                       uint32_t* hist, uint32_t* scan_tmp) {
    if (!cap) return;
    if (key16) radix_pass_t<uint16_t>(s, (const uint16_t*)keys_in, vals_in, msk_in, (uint16_t*)keys_out, vals_out, msk_out, pairs_total, cap, shift, bits, hist, scan_tmp);
    else radix_pass_t<uint32_t>(s, (const uint32_t*)keys_in, vals_in, msk_in, (uint32_t*)keys_out, vals_out, msk_out, pairs_total, cap, shift, bits, hist, scan_tmp);
}

void launch_bounds(hipStream_t s, const void* keys, bool key16, const unsigned long long* pairs_total, uint32_t cap,
                   uint32_t* tile_start, uint32_t* tile_end) {
    if (!cap) return;
    if (key16) hipLaunchKernelGGL(k_bounds<uint16_t>, dim3((unsigned)(((uint64_t)cap + 1023) / 1024)), dim3(256), 0, s, (const uint16_t*)keys, pairs_total, cap, tile_start, tile_end);
    else hipLaunchKernelGGL(k_bounds<uint32_t>, dim3((unsigned)(((uint64_t)cap + 1023) / 1024)), dim3(256), 0, s, (const uint32_t*)keys, pairs_total, cap, tile_start, tile_end);
}

}  // namespace trgl
