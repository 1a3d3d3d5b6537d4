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

// one thread per face-vertex
__global__ __launch_bounds__(256) void k_vertex_stage(VertexStageParams p) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)p.nfaces * 3) return;
    const uint64_t f = i / 3; const int v = (int)(i - f * 3);
    const double* vert = p.vertices + (size_t)p.indices[i] * p.stride;       // model.cpp:396-412
    const double px = vert[0], py = vert[1], pz = vert[2], nx = vert[3], ny = vert[4], nz = vert[5];
    double eye[4], nrm[3];
#pragma unroll
    for (int r = 0; r < 4; ++r) eye[r] = dot4(p.mv + 4 * r, px, py, pz, 1.0);                 // main.cpp:77-80
#pragma unroll
    for (int r = 0; r < 3; ++r) nrm[r] = dot4(p.mv + 4 * r, nx, ny, nz, 0.0);                 // main.cpp:83-86
    double* uv = p.vary + 24 * f;
    uv[2 * v] = vert[6]; uv[2 * v + 1] = vert[7];                                              // main.cpp:75
#pragma unroll
    for (int k = 0; k < 3; ++k) { uv[6 + 3 * v + k] = eye[k]; uv[15 + 3 * v + k] = nrm[k]; }  // main.cpp:81,87
#pragma unroll
    for (int r = 0; r < 4; ++r) p.clip[12 * f + 4 * v + r] = dot4(p.proj + 4 * r, eye[0], eye[1], eye[2], eye[3]);  // :89
}

// ---- N4 ------------------------------------------------------------------------------------------------------
// min / max over the finite depths (main.cpp:275-281); keys[0] = min key, keys[1] = max key
__global__ __launch_bounds__(256) void k_zrange(const double* __restrict__ zb, uint64_t n, unsigned long long* __restrict__ keys) {
    unsigned long long kmin = ~0ull, kmax = 0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const double d = zb[i];
        if (__builtin_isfinite(d)) { const unsigned long long k = zkey(d); kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax; }
    }
    for (int o = 32; o; o >>= 1) {
        unsigned long long a = __shfl_xor(kmin, o); kmin = a < kmin ? a : kmin;
        unsigned long long b = __shfl_xor(kmax, o); kmax = b > kmax ? b : kmax;
    }
    if ((threadIdx.x & 63) == 0) {
        if (kmin != ~0ull) atomicMin(&keys[0], kmin);
        if (kmax != 0ull) atomicMax(&keys[1], kmax);
    }
}

__global__ __launch_bounds__(256) void k_zimage(const double* __restrict__ zb, uint64_t n, const unsigned long long* __restrict__ keys,
                                                uint8_t* __restrict__ out) {
    // main.cpp:275: min_depth = 1e9, max_depth = -1e9 before the scan (so an all-empty buffer keeps them)
    double min_depth = 1e9, max_depth = -1e9;
    if (keys[0] != ~0ull) { min_depth = dmin(min_depth, zkey_decode(keys[0])); max_depth = dmax(max_depth, zkey_decode(keys[1])); }
    if (max_depth - min_depth < 1e-7) max_depth = min_depth + 1e-7;                  // :294-296
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double depth = zb[i];
    unsigned char value = 255;
    if (__builtin_isfinite(depth)) {
        const double normalized = (depth - min_depth) / (max_depth - min_depth);    // :305
        value = (unsigned char)(255.0 * (1.0 - normalized));                        // :306
    }
    out[3 * i] = value; out[3 * i + 1] = value; out[3 * i + 2] = value;
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
    for (int k = threadIdx.x; k < SS_LDS * SS_LDS; k += 256) {
        const int lx = k % SS_LDS, ly = k / SS_LDS;
        const int gx = bx0 - SS_HALO + lx, gy = by0 - SS_HALO + ly;
        s_z[k] = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? zb[(size_t)gx + (size_t)gy * W] : 0.0;
    }
    __syncthreads();
    for (int q = 0; q < 4; ++q) {
        const int lx = threadIdx.x & 31, ly = (threadIdx.x >> 5) + 8 * q;
        const int pixel_x = bx0 + lx, pixel_y = by0 + ly;
        if (pixel_x >= W || pixel_y >= H) continue;
        const double center_depth = s_z[(ly + SS_HALO) * SS_LDS + lx + SS_HALO];
        double ao_value = 1.0;
        if (__builtin_isfinite(center_depth)) {                                              // main.cpp:328
            int occluded = 0, total = 0;
            for (int d = 0; d < sp.num_directions; ++d) {
                const double dir_x = sp.dir_x[d], dir_y = sp.dir_y[d];
                for (int step = 1; step <= sp.steps; ++step) {
                    const double radius = (double)step / sp.steps * sp.sample_radius;        // :337
                    const int sample_x = (int)round(pixel_x + dir_x * radius);               // :338-339
                    const int sample_y = (int)round(pixel_y + dir_y * radius);
                    if (sample_x < 0 || sample_x >= W || sample_y < 0 || sample_y >= H) continue;
                    const double sample_depth = lds_ok
                        ? s_z[(sample_y - by0 + SS_HALO) * SS_LDS + (sample_x - bx0 + SS_HALO)]
                        : zb[(size_t)sample_x + (size_t)sample_y * W];
                    if (!__builtin_isfinite(sample_depth)) { total++; continue; }            // :346-349
                    if (sample_depth < center_depth - sp.threshold) occluded++;              // :351-353
                    total++;
                }
            }
            if (total != 0) {
                const double occlusion_factor = (double)occluded / (double)total;            // :360
                ao_value = 1.0 - occlusion_factor * sp.intensity;                            // :361
            }
        }
        const unsigned char intensity = (unsigned char)(255.0 * ao_value);                   // :760
        const size_t i = (size_t)pixel_x + (size_t)pixel_y * W;
        out[3 * i] = intensity; out[3 * i + 1] = intensity; out[3 * i + 2] = intensity;
    }
}

__global__ __launch_bounds__(256) void k_composite(const uint8_t* __restrict__ fb, int bpp, const uint8_t* __restrict__ ao, uint64_t n,
                                                   uint8_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double ao_factor = ao[3 * i] / 255.0;                                              // main.cpp:775
    for (int c = 0; c < 3; ++c)
        out[3 * i + c] = (unsigned char)dmin(255.0, (double)fb[i * bpp + c] * ao_factor);    // :777-781
}

}  // namespace

namespace trgl {

void launch_vertex_stage(hipStream_t s, const double mv[16], const double proj[16], const double* vertices, int stride,
                         const uint32_t* indices, uint32_t nfaces, double* clip, double* vary) {
    if (!nfaces) return;
    VertexStageParams p;
    for (int i = 0; i < 16; ++i) { p.mv[i] = mv[i]; p.proj[i] = proj[i]; }
    p.vertices = vertices; p.indices = indices; p.clip = clip; p.vary = vary; p.nfaces = nfaces; p.stride = stride;
    const uint64_t n = (uint64_t)nfaces * 3;
    hipLaunchKernelGGL(k_vertex_stage, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p);
}

void launch_zimage(hipStream_t s, const double* zb, int W, int H, unsigned long long* keys2, uint8_t* out) {
    const uint64_t n = (uint64_t)W * H;
    static const unsigned long long init[2] = { ~0ull, 0ull };
    (void)hipMemcpyAsync(keys2, init, 16, hipMemcpyHostToDevice, s);
    hipLaunchKernelGGL(k_zrange, dim3(1024), dim3(256), 0, s, zb, n, keys2);
    hipLaunchKernelGGL(k_zimage, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, zb, n, keys2, out);
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
    hipLaunchKernelGGL(k_composite, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fb, bpp, ao, n, out);
}

}  // namespace trgl
