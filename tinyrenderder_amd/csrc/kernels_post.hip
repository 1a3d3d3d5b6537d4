// kernels_post.hip — the two callers either side of the rasterizer, gfx950.
//
//   N1  k_vertex_stage : PhongShader::vertex / EyeShader::vertex (main.cpp:71-90,199-218) for an indexed mesh:
//                        eye = MV*(p,1), normal_eye = MV*(n,0), clip = P*eye; emits the clip / varyings arrays the
//                        rasterizer consumes, so the host face loop and 288 B/triangle of H2D disappear.
//   N4  k_zrange + k_zimage : save_zbuffer_image's pixels (main.cpp:269-311)
//       k_ssao              : compute_ssao_at for every pixel (main.cpp:317-362,757-763); the 33x33-pixel
//                             neighbourhood of a 32x32 block comes from one 64x64 fp64 tile in LDS
//       k_composite         : final = phong * ao (main.cpp:768-783)
//   They work on the z-buffer / framebuffer already resident in HBM: no z-buffer D2H for the post-process.
// fp64 in the reference's operation order, contraction off.
#include <hip/hip_runtime.h>
#include "trgl_device.h"
#include "launch.h"

namespace {

__device__ __forceinline__ double dot4(const double* m, double x, double y, double z, double w) {
    double sum = 0;                                   // geometry.h:122-127
    sum += m[0] * x; sum += m[1] * y; sum += m[2] * z; sum += m[3] * w;
    return sum;
}
__device__ __forceinline__ double dmax(double a, double b) { return (a < b) ? b : a; }
__device__ __forceinline__ double dmin(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ unsigned long long zkey(double d) {
    unsigned long long b = (unsigned long long)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double zkey_decode(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

struct VertexStageParams {
    double mv[16], proj[16];
    const double* vertices; const uint32_t* indices;
    double* clip; double* vary;
    uint32_t nfaces; int32_t stride;
};

// One thread per face-vertex; a block is 64 faces (192 face-vertices) whose 96-byte clip rows and 192-byte varyings rows are assembled
// in LDS and leave as contiguous 16-byte stores (twelve scattered 8-byte stores per thread before: 57 -> 45 us for the 327 k-face head
// with a shuffled index buffer; the rest is the random 64-byte gather of the vertices).
constexpr int VS_FACES = 64;
__global__ __launch_bounds__(VS_FACES * 3) void k_vertex_stage(VertexStageParams p) {
    __shared__ __attribute__((aligned(16))) double s_clip[VS_FACES * 12];
    __shared__ __attribute__((aligned(16))) double s_vary[VS_FACES * 24];
    const uint64_t f0 = (uint64_t)blockIdx.x * VS_FACES;
    const uint64_t i = f0 * 3 + threadIdx.x;
    const uint32_t nf_blk = (uint32_t)(p.nfaces - f0 < (uint64_t)VS_FACES ? p.nfaces - f0 : VS_FACES);
    if (i < (uint64_t)p.nfaces * 3) {
        const int f = threadIdx.x / 3, v = threadIdx.x - 3 * f;
        const double* vert = p.vertices + (size_t)p.indices[i] * p.stride;       // model.cpp:396-412
        const double px = vert[0], py = vert[1], pz = vert[2], nx = vert[3], ny = vert[4], nz = vert[5];
        double eye[4], nrm[3];
#pragma unroll
        for (int r = 0; r < 4; ++r) eye[r] = dot4(p.mv + 4 * r, px, py, pz, 1.0);                 // main.cpp:77-80
#pragma unroll
        for (int r = 0; r < 3; ++r) nrm[r] = dot4(p.mv + 4 * r, nx, ny, nz, 0.0);                 // main.cpp:83-86
        double* uv = s_vary + 24 * f;
        uv[2 * v] = vert[6]; uv[2 * v + 1] = vert[7];                                              // main.cpp:75
#pragma unroll
        for (int k = 0; k < 3; ++k) { uv[6 + 3 * v + k] = eye[k]; uv[15 + 3 * v + k] = nrm[k]; }  // main.cpp:81,87
#pragma unroll
        for (int r = 0; r < 4; ++r) s_clip[12 * f + 4 * v + r] = dot4(p.proj + 4 * r, eye[0], eye[1], eye[2], eye[3]);  // :89
    }
    __syncthreads();
    {
        const double2* sc = reinterpret_cast<const double2*>(s_clip);
        double2* gc = reinterpret_cast<double2*>(p.clip + 12 * f0);
        for (uint32_t k = threadIdx.x; k < nf_blk * 6; k += VS_FACES * 3) gc[k] = sc[k];
        const double2* sv = reinterpret_cast<const double2*>(s_vary);
        double2* gv = reinterpret_cast<double2*>(p.vary + 24 * f0);
        for (uint32_t k = threadIdx.x; k < nf_blk * 12; k += VS_FACES * 3) gv[k] = sv[k];
    }
}

// ---- N4 ------------------------------------------------------------------------------------------------------
// min / max over the finite depths (main.cpp:275-281); keys[0] = min key, keys[1] = max key
__global__ __launch_bounds__(256) void k_zrange(const double* __restrict__ zb, uint64_t n, unsigned long long* __restrict__ keys) {
    unsigned long long kmin = ~0ull, kmax = 0ull;
    // eight 16-byte loads in flight per thread (one load per iteration left the kernel waiting on memory: 1.5 TB/s)
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n2 = n / 2;                                    // pairs of depths (the z-buffer is 16-byte aligned)
    const double2* zb2 = reinterpret_cast<const double2*>(zb);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += 8 * stride) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const uint64_t j = i + u * stride; v[u] = j < n2 ? zb2[j] : make_double2(__builtin_inf(), __builtin_inf()); }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double d2[2] = { v[u].x, v[u].y };
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (__builtin_isfinite(d2[e])) { const unsigned long long k = zkey(d2[e]); kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax; }
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const double d = zb[n - 1];
        if (__builtin_isfinite(d)) { const unsigned long long k = zkey(d); kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax; }
    }
    for (int o = 32; o; o >>= 1) {
        unsigned long long a = __shfl_xor(kmin, o); kmin = a < kmin ? a : kmin;
        unsigned long long b = __shfl_xor(kmax, o); kmax = b > kmax ? b : kmax;
    }
    // one pair of atomics per BLOCK, and only when it can change the value: same-address atomics serialise at ~10 ns each, and
    // 8192 of them (one pair per wave) were most of this kernel's 90 us.  (A stale plain load can cause a redundant atomic, never a missed one.)
    __shared__ unsigned long long s_min[4], s_max[4];
    if ((threadIdx.x & 63) == 0) { s_min[threadIdx.x >> 6] = kmin; s_max[threadIdx.x >> 6] = kmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { kmin = s_min[w] < kmin ? s_min[w] : kmin; kmax = s_max[w] > kmax ? s_max[w] : kmax; }
        if (kmin != ~0ull && kmin < __builtin_nontemporal_load(&keys[0])) atomicMin(&keys[0], kmin);
        if (kmax != 0ull && kmax > __builtin_nontemporal_load(&keys[1])) atomicMax(&keys[1], kmax);
    }
}

__global__ __launch_bounds__(256) void k_zimage(const double* __restrict__ zb, uint64_t n, const unsigned long long* __restrict__ keys,
                                                uint8_t* __restrict__ out) {
    // main.cpp:275: min_depth = 1e9, max_depth = -1e9 before the scan (so an all-empty buffer keeps them)
    double min_depth = 1e9, max_depth = -1e9;
    if (keys[0] != ~0ull) { min_depth = dmin(min_depth, zkey_decode(keys[0])); max_depth = dmax(max_depth, zkey_decode(keys[1])); }
    if (max_depth - min_depth < 1e-7) max_depth = min_depth + 1e-7;                  // :294-296
    // four pixels per thread: two 16-byte loads, twelve output bytes as three 4-byte stores (byte stores were most of the instructions)
    const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    double d[4];
    if (i0 + 4 <= n) {
        const double2 a = *reinterpret_cast<const double2*>(zb + i0), b2 = *reinterpret_cast<const double2*>(zb + i0 + 2);
        d[0] = a.x; d[1] = a.y; d[2] = b2.x; d[3] = b2.y;
    } else {
        for (int j = 0; j < 4; ++j) d[j] = i0 + j < n ? zb[i0 + j] : __builtin_inf();
    }
    uint32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned char value = 255;
        if (__builtin_isfinite(d[j])) {
            const double normalized = (d[j] - min_depth) / (max_depth - min_depth);    // :305
            value = (unsigned char)(255.0 * (1.0 - normalized));                        // :306
        }
        v[j] = value;
    }
    if (i0 + 4 <= n) {       // 12 bytes at a multiple of 12: three aligned words  v0 v0 v0 v1 | v1 v1 v2 v2 | v2 v3 v3 v3
        uint32_t* o = reinterpret_cast<uint32_t*>(out + 3 * i0);
        o[0] = v[0] * 0x010101u | (v[1] << 24);
        o[1] = v[1] * 0x0101u | (v[2] << 16) | (v[2] << 24);
        o[2] = v[2] | (v[3] * 0x010101u << 8);
    } else {
        for (int j = 0; j < 4 && i0 + j < n; ++j) { out[3 * (i0 + j)] = (uint8_t)v[j]; out[3 * (i0 + j) + 1] = (uint8_t)v[j]; out[3 * (i0 + j) + 2] = (uint8_t)v[j]; }
    }
}

struct SsaoParams {
    double dir_x[16], dir_y[16];     // cos / sin of 2*pi*d/N, evaluated by the host's libm as the reference does
    int32_t num_directions, steps;
    double sample_radius, threshold, intensity;
};

constexpr int SS_TILE = 32, SS_HALO = 16, SS_LDS = SS_TILE + 2 * SS_HALO;     // 64 x 64 doubles = 32 KB

__global__ __launch_bounds__(256) void k_ssao(const double* __restrict__ zb, int W, int H, SsaoParams sp, uint8_t* __restrict__ out) {
    __shared__ double s_z[SS_LDS * SS_LDS];
    const int bx0 = blockIdx.x * SS_TILE, by0 = blockIdx.y * SS_TILE;
    const bool lds_ok = sp.sample_radius <= (double)SS_HALO;      // larger radii read the z-buffer directly
    // The block's own 32x32 depths first: a block without a finite one (background: three quarters of a mesh frame) writes ao = 1.0
    // and is done without ever loading the 16-pixel halo.
    // A depth that is not finite is only ever tested with isfinite(): +inf stands for all of them in the tile (and for NaN and -inf
    // `sample < limit` is false, which the interior path below relies on).
    int any_finite = 0;
    for (int k = threadIdx.x; k < SS_TILE * SS_TILE; k += 256) {
        const int lx = k % SS_TILE, ly = k / SS_TILE;
        const int gx = bx0 + lx, gy = by0 + ly;
        double zv = (gx < W && gy < H) ? zb[(size_t)gx + (size_t)gy * W] : __builtin_inf();
        if (!__builtin_isfinite(zv)) zv = __builtin_inf(); else any_finite = 1;
        s_z[(ly + SS_HALO) * SS_LDS + lx + SS_HALO] = zv;
    }
    if (!__syncthreads_or(any_finite)) {
        for (int k = threadIdx.x; k < SS_TILE * SS_TILE; k += 256) {
            const int gx = bx0 + k % SS_TILE, gy = by0 + k / SS_TILE;
            if (gx < W && gy < H) { const size_t i = (size_t)gx + (size_t)gy * W; out[3 * i] = 255; out[3 * i + 1] = 255; out[3 * i + 2] = 255; }   // (unsigned char)(255.0 * 1.0), :760
        }
        return;
    }
    for (int k = threadIdx.x; k < SS_LDS * SS_LDS; k += 256) {
        const int lx = k % SS_LDS, ly = k / SS_LDS;
        if (lx >= SS_HALO && lx < SS_HALO + SS_TILE && ly >= SS_HALO && ly < SS_HALO + SS_TILE) continue;     // loaded above
        const int gx = bx0 - SS_HALO + lx, gy = by0 - SS_HALO + ly;
        double zv = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? zb[(size_t)gx + (size_t)gy * W] : 0.0;
        if (!__builtin_isfinite(zv)) zv = __builtin_inf();
        s_z[k] = zv;
    }
    __syncthreads();
    // Sample offsets as INTEGERS when that is exact.  (int)round(pixel + t) with t = dir * radius the same for every pixel is
    // pixel + o for one integer o, unless the fp64 sum pixel + t rounds across a tie for some pixel of the block; each sample k =
    // (direction, step) is checked against the literal expression on the 32 columns and 32 rows of the block (2 x 32 evaluations per
    // sample instead of 1024), and a block with any disagreement, or with more samples than the table holds, takes the literal path.
    // The integer path costs an add where the literal one costs an fp64 add, a round and a conversion, twice per sample.
    constexpr int SS_MAXK = 256;
    __shared__ int s_ox[SS_MAXK], s_oy[SS_MAXK];
    __shared__ int s_bad;
    const int K = sp.num_directions * sp.steps;
    if (threadIdx.x == 0) s_bad = (K > SS_MAXK) ? 1 : 0;
    __syncthreads();
    if (K <= SS_MAXK) {
        for (int k = threadIdx.x; k < K; k += 256) {
            const int d = k / sp.steps, step = k - d * sp.steps + 1;
            const double radius = (double)step / sp.steps * sp.sample_radius;                // :337
            const double tx = sp.dir_x[d] * radius, ty = sp.dir_y[d] * radius;
            const int ox = (int)round(bx0 + tx) - bx0, oy = (int)round(by0 + ty) - by0;      // :338-339 at the block's first column / row
            bool same = true;
            for (int j = 1; j < SS_TILE; ++j) {
                same = same && ((int)round((bx0 + j) + tx) - (bx0 + j) == ox) && ((int)round((by0 + j) + ty) - (by0 + j) == oy);
            }
            s_ox[k] = ox; s_oy[k] = oy;
            if (!same) s_bad = 1;
        }
    }
    __syncthreads();
    const bool int_path = s_bad == 0;

    // A thread owns the four pixels (lx, ly + 8 q) of its column: the sample's x is the same for all four, so the sample loop is
    // outermost and the four pixels innermost.
    const int lx = threadIdx.x & 31, ly0 = threadIdx.x >> 5;
    const int pixel_x = bx0 + lx;
    if (pixel_x >= W) return;
    double center_depth[4]; bool live[4]; int occluded[4], total[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ly = ly0 + 8 * q;
        center_depth[q] = s_z[(ly + SS_HALO) * SS_LDS + lx + SS_HALO];
        live[q] = by0 + ly < H && __builtin_isfinite(center_depth[q]);                       // main.cpp:328
        occluded[q] = 0; total[q] = 0;
    }
    // a wave without a finite centre (background) has nothing to sample: ao = 1.0 for its pixels
    const bool wave_live = __ballot(live[0] || live[1] || live[2] || live[3]) != 0;
    const bool interior = bx0 >= SS_HALO && bx0 + SS_TILE + SS_HALO <= W && by0 >= SS_HALO && by0 + SS_TILE + SS_HALO <= H;
    if (!wave_live) {
    } else if (int_path && lds_ok && interior) {
        // No sample of this block leaves the image: every live pixel counts all K samples (:355), and a sample is occluding iff its
        // depth is below the limit (non-finite depths sit in the tile as +inf): one LDS read, one compare, one add per sample.
        double limit[4]; int row_base[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            limit[q] = center_depth[q] - sp.threshold;                                       // the right-hand side of :351
            row_base[q] = (ly0 + 8 * q + SS_HALO) * SS_LDS + lx + SS_HALO;
            total[q] = live[q] ? K : 0;
        }
        for (int k = 0; k < K; ++k) {
            const int off = s_oy[k] * SS_LDS + s_ox[k];
#pragma unroll
            for (int q = 0; q < 4; ++q) occluded[q] += (s_z[row_base[q] + off] < limit[q]) ? 1 : 0;   // :351-353
        }
    } else if (int_path && lds_ok) {
        // every sample lies in the LDS tile (|offset| <= 16 = the halo), so it is read unconditionally and counted by predicates:
        // no branch per sample and pixel
        double limit[4]; int row_base[4], py[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            limit[q] = center_depth[q] - sp.threshold;                                       // the right-hand side of :351
            py[q] = by0 + ly0 + 8 * q;
            row_base[q] = (ly0 + 8 * q + SS_HALO) * SS_LDS + lx + SS_HALO;
        }
        for (int k = 0; k < K; ++k) {
            const int ox = s_ox[k], oy = s_oy[k];
            const int sample_x = pixel_x + ox;
            const bool x_in = sample_x >= 0 && sample_x < W;                                 // :340
            const int off = oy * SS_LDS + ox;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int sample_y = py[q] + oy;
                const double sample_depth = s_z[row_base[q] + off];
                const bool in = live[q] && x_in && sample_y >= 0 && sample_y < H;
                total[q] += in ? 1 : 0;                                                      // :346-349, :355: counted finite or not
                occluded[q] += (in && __builtin_isfinite(sample_depth) && sample_depth < limit[q]) ? 1 : 0;   // :351-353
            }
        }
    } else if (int_path) {
        for (int k = 0; k < K; ++k) {
            const int sample_x = pixel_x + s_ox[k], oy = s_oy[k];
            if (sample_x < 0 || sample_x >= W) continue;                                     // (the x half of :340)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!live[q]) continue;
                const int sample_y = by0 + ly0 + 8 * q + oy;
                if (sample_y < 0 || sample_y >= H) continue;
                const double sample_depth = zb[(size_t)sample_x + (size_t)sample_y * W];
                if (!__builtin_isfinite(sample_depth)) { total[q]++; continue; }             // :346-349
                if (sample_depth < center_depth[q] - sp.threshold) occluded[q]++;            // :351-353
                total[q]++;
            }
        }
    } else {
        for (int d = 0; d < sp.num_directions; ++d) {
            const double dir_x = sp.dir_x[d], dir_y = sp.dir_y[d];
            for (int step = 1; step <= sp.steps; ++step) {
                const double radius = (double)step / sp.steps * sp.sample_radius;            // :337
                const int sample_x = (int)round(pixel_x + dir_x * radius);                   // :338
                if (sample_x < 0 || sample_x >= W) continue;                                 // (the x half of :340)
                const double ry = dir_y * radius;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (!live[q]) continue;
                    const int pixel_y = by0 + ly0 + 8 * q;
                    const int sample_y = (int)round(pixel_y + ry);                           // :339
                    if (sample_y < 0 || sample_y >= H) continue;
                    const double sample_depth = lds_ok
                        ? s_z[(sample_y - by0 + SS_HALO) * SS_LDS + (sample_x - bx0 + SS_HALO)]
                        : zb[(size_t)sample_x + (size_t)sample_y * W];
                    if (!__builtin_isfinite(sample_depth)) { total[q]++; continue; }         // :346-349
                    if (sample_depth < center_depth[q] - sp.threshold) occluded[q]++;        // :351-353
                    total[q]++;
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pixel_y = by0 + ly0 + 8 * q;
        if (pixel_y >= H) continue;
        double ao_value = 1.0;
        if (live[q] && total[q] != 0) {
            const double occlusion_factor = (double)occluded[q] / (double)total[q];          // :360
            ao_value = 1.0 - occlusion_factor * sp.intensity;                                // :361
        }
        const unsigned char intensity = (unsigned char)(255.0 * ao_value);                   // :760
        const size_t i = (size_t)pixel_x + (size_t)pixel_y * W;
        out[3 * i] = intensity; out[3 * i + 1] = intensity; out[3 * i + 2] = intensity;
    }
}

__global__ __launch_bounds__(256) void k_composite(const uint8_t* __restrict__ fb, int bpp, const uint8_t* __restrict__ ao, uint64_t n,
                                                   uint8_t* __restrict__ out) {
    const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    if (bpp == 3 && i0 + 4 <= n) {       // four RGB pixels: three aligned words in, three out
        const uint32_t* f = reinterpret_cast<const uint32_t*>(fb + 3 * i0);
        const uint32_t* a = reinterpret_cast<const uint32_t*>(ao + 3 * i0);
        const uint32_t fw[3] = { f[0], f[1], f[2] }, aw[3] = { a[0], a[1], a[2] };
        uint32_t ow[3] = { 0, 0, 0 };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ab = 3 * j;                                                            // byte of the pixel's first ao channel
            const double ao_factor = (double)((aw[ab >> 2] >> (8 * (ab & 3))) & 0xffu) / 255.0;   // main.cpp:775
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int bi = 3 * j + c;
                const double ch = (double)((fw[bi >> 2] >> (8 * (bi & 3))) & 0xffu);
                ow[bi >> 2] |= (uint32_t)(unsigned char)dmin(255.0, ch * ao_factor) << (8 * (bi & 3));   // :777-781
            }
        }
        uint32_t* o = reinterpret_cast<uint32_t*>(out + 3 * i0);
        o[0] = ow[0]; o[1] = ow[1]; o[2] = ow[2];
        return;
    }
    for (uint64_t i = i0; i < i0 + 4 && i < n; ++i) {
        const double ao_factor = ao[3 * i] / 255.0;                                          // main.cpp:775
        for (int c = 0; c < 3; ++c)
            out[3 * i + c] = (unsigned char)dmin(255.0, (double)fb[i * bpp + c] * ao_factor);    // :777-781
    }
}

}  // namespace

namespace trgl {

void launch_vertex_stage(hipStream_t s, const double mv[16], const double proj[16], const double* vertices, int stride,
                         const uint32_t* indices, uint32_t nfaces, double* clip, double* vary) {
    if (!nfaces) return;
    VertexStageParams p;
    for (int i = 0; i < 16; ++i) { p.mv[i] = mv[i]; p.proj[i] = proj[i]; }
    p.vertices = vertices; p.indices = indices; p.clip = clip; p.vary = vary; p.nfaces = nfaces; p.stride = stride;
    hipLaunchKernelGGL(k_vertex_stage, dim3((nfaces + VS_FACES - 1) / VS_FACES), dim3(VS_FACES * 3), 0, s, p);
}

void launch_zimage(hipStream_t s, const double* zb, int W, int H, unsigned long long* keys2, uint8_t* out) {
    const uint64_t n = (uint64_t)W * H;
    static const unsigned long long init[2] = { ~0ull, 0ull };
    (void)hipMemcpyAsync(keys2, init, 16, hipMemcpyHostToDevice, s);
    hipLaunchKernelGGL(k_zrange, dim3(1024), dim3(256), 0, s, zb, n, keys2);
    hipLaunchKernelGGL(k_zimage, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, zb, n, keys2, out);
}

void launch_ssao(hipStream_t s, const double* zb, int W, int H, const double* dir_x, const double* dir_y, int ndir, int steps,
                 double radius, double threshold, double intensity, uint8_t* out) {
    SsaoParams sp;
    for (int i = 0; i < 16; ++i) { sp.dir_x[i] = i < ndir ? dir_x[i] : 0.0; sp.dir_y[i] = i < ndir ? dir_y[i] : 0.0; }
    sp.num_directions = ndir; sp.steps = steps; sp.sample_radius = radius; sp.threshold = threshold; sp.intensity = intensity;
    hipLaunchKernelGGL(k_ssao, dim3((W + SS_TILE - 1) / SS_TILE, (H + SS_TILE - 1) / SS_TILE), dim3(256), 0, s, zb, W, H, sp, out);
}

void launch_composite(hipStream_t s, const uint8_t* fb, int bpp, const uint8_t* ao, int W, int H, uint8_t* out) {
    const uint64_t n = (uint64_t)W * H;
    hipLaunchKernelGGL(k_composite, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, fb, bpp, ao, n, out);
}

}  // namespace trgl
