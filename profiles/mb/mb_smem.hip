// Micro-benchmark (round 3): how fast can waves stream 128-B triangle records as wave-uniform constants?
//   mode 0: no record loads (pure VALU chain of `nops` fp64 FMAs per visit)            -> VALU floor
//   mode 1: scalar loads (s_load_dwordx16 x2 per visit) from recs[idx], one visit ahead -> SMEM / K$ streaming rate
//   mode 2: LDS broadcast: lane (v&63) of the wave loaded the record, writes 128 B to a slot, all lanes read it back
// Each wave walks `visits` indices (a per-wave slice of a random index array; every record is used ~`reuse` times
// by waves of the same workgroup, as the blocks of one tile would).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
struct alignas(128) Rec { double d[16]; };
typedef const __attribute__((address_space(4))) Rec CRec;

template <int MODE, int NOPS>
__global__ __launch_bounds__(256) void k(const Rec* __restrict__ recs, const uint32_t* __restrict__ idx, int visits, double* __restrict__ out, int lds_pad) {
    extern __shared__ double pad[];
    __shared__ __attribute__((aligned(16))) double slot[4][2][16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t* my = idx + (size_t)(blockIdx.x * 4 + w) * visits;
    double acc0 = lane, acc1 = 1.0 + lane;
    if (lds_pad > 1 << 30) acc0 += pad[lane];
    uint32_t cur_i = __builtin_amdgcn_readfirstlane(my[0]);
    if (MODE == 1) {
#if __HIP_DEVICE_COMPILE__
        CRec* cr = (CRec*)recs;
        double cur[16], nxt[16];
        for (int q = 0; q < 16; ++q) cur[q] = cr[cur_i].d[q];
        for (int v = 0; v < visits; ++v) {
            const uint32_t nxt_i = __builtin_amdgcn_readfirstlane(my[v + 1 < visits ? v + 1 : v]);
            for (int q = 0; q < 16; ++q) nxt[q] = cr[nxt_i].d[q];
#pragma unroll
            for (int o = 0; o < NOPS; ++o) {
                acc0 = __builtin_fma(acc0, cur[o & 15], cur[(o + 5) & 15]);
                acc1 = __builtin_fma(acc1, cur[(o + 3) & 15], cur[(o + 9) & 15]);
            }
            for (int q = 0; q < 16; ++q) cur[q] = nxt[q];
        }
#endif
    } else if (MODE == 3) {
#if __HIP_DEVICE_COMPILE__
        typedef uint32_t u16v __attribute__((ext_vector_type(16)));
        typedef double d8v __attribute__((ext_vector_type(8)));
        auto ld = [&](uint32_t i, u16v& lo, u16v& hi) {
            const Rec* p = recs + i;
            asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(lo), "=&s"(hi) : "s"(p));
        };
        auto use = [&](const u16v& lo, const u16v& hi) {
            const d8v a = __builtin_bit_cast(d8v, lo), b = __builtin_bit_cast(d8v, hi);
            double c[16];
            for (int q = 0; q < 8; ++q) { c[q] = a[q]; c[8 + q] = b[q]; }
#pragma unroll
            for (int o = 0; o < NOPS; ++o) {
                acc0 = __builtin_fma(acc0, c[o & 15], acc1);
                acc1 = __builtin_fma(acc1, c[(o + 3) & 15], acc0);
            }
        };
        u16v alo, ahi, blo, bhi;
        ld(cur_i, alo, ahi);
        for (int v = 0; v < visits; v += 2) {
            const uint32_t i1 = __builtin_amdgcn_readfirstlane(my[v + 1 < visits ? v + 1 : v]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+s"(alo), "+s"(ahi));
            ld(i1, blo, bhi);
            use(alo, ahi);
            const uint32_t i2 = __builtin_amdgcn_readfirstlane(my[v + 2 < visits ? v + 2 : v]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+s"(blo), "+s"(bhi));
            ld(i2, alo, ahi);
            use(blo, bhi);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    } else if (MODE == 2) {
        for (int v = 0; v < visits; v += 1) {
            const uint32_t i = __builtin_amdgcn_readfirstlane(my[v]);
            // one lane "owns" the record (as the survivor lane of a batch would) and publishes it
            if (lane == (v & 63)) {
                const double2* p = (const double2*)&recs[i];
#pragma unroll
                for (int c = 0; c < 8; ++c) ((double2*)slot[w][v & 1])[c] = p[c];
            }
            __builtin_amdgcn_wave_barrier();
            double c[16];
#pragma unroll
            for (int q = 0; q < 8; ++q) { const double2 t = ((const double2*)slot[w][v & 1])[q]; c[2 * q] = t.x; c[2 * q + 1] = t.y; }
#pragma unroll
            for (int o = 0; o < NOPS; ++o) {
                acc0 = __builtin_fma(acc0, c[o & 15], c[(o + 5) & 15]);
                acc1 = __builtin_fma(acc1, c[(o + 3) & 15], c[(o + 9) & 15]);
            }
        }
    } else {
        double c[16];
        for (int q = 0; q < 16; ++q) c[q] = 1.0 + 1e-9 * q;
        for (int v = 0; v < visits; ++v) {
#pragma unroll
            for (int o = 0; o < NOPS; ++o) {
                acc0 = __builtin_fma(acc0, c[o & 15], c[(o + 5) & 15]);
                acc1 = __builtin_fma(acc1, c[(o + 3) & 15], c[(o + 9) & 15]);
            }
            asm volatile("" : "+v"(acc0), "+v"(acc1));
        }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc0 + acc1;
}

template <int MODE, int NOPS>
void run(const char* name, const Rec* recs, const uint32_t* idx, int nblocks, int visits, double* out, int lds_bytes) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<MODE, NOPS>), dim3(nblocks), dim3(256), lds_bytes, 0, recs, idx, visits, out, lds_bytes);
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k<MODE, NOPS>), dim3(nblocks), dim3(256), lds_bytes, 0, recs, idx, visits, out, lds_bytes);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    const double nvis = (double)nblocks * 4 * visits;
    printf("%-34s lds/block %6d B: %8.3f ms  %7.1f M visits/ms-chip  %7.2f TB/s of records  %6.1f VALU-cycles/visit/SIMD\n", name, lds_bytes, ms,
           nvis / ms * 1e-6 * 1e3 / 1e3, nvis * 128 / ms * 1e-9, ms * 1e-3 * 2.4e9 * 1024 / nvis);
}

int main() {
    const size_t nrec = 10u << 20;            // 1.34 GB of records
    const int visits = 80, nblocks = 65536;   // 262144 waves x 80 visits = 21 M visits
    Rec* recs; CK(hipMalloc(&recs, nrec * sizeof(Rec)));
    CK(hipMemset(recs, 0, nrec * sizeof(Rec)));
    std::vector<uint32_t> h((size_t)nblocks * 4 * visits);
    uint64_t st = 12345;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(st >> 33); };
    // the 4 waves of a block share a pool of `visits*2` records (each record used by ~2 waves), pools are random over the array
    for (int b = 0; b < nblocks; ++b) {
        std::vector<uint32_t> pool(visits * 2);
        for (auto& p : pool) p = rnd() % nrec;
        for (int w = 0; w < 4; ++w) for (int v = 0; v < visits; ++v) h[((size_t)b * 4 + w) * visits + v] = pool[rnd() % pool.size()];
    }
    uint32_t* idx; CK(hipMalloc(&idx, h.size() * 4)); CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    double* out; CK(hipMalloc(&out, (size_t)nblocks * 256 * 8));
    for (int lds : {0, 20000, 40000}) {       // 0: up to 8 blocks/CU (VGPR permitting); 20000: 8 blocks; 40000: 4 blocks/CU = 4 waves/SIMD
        run<0, 16>("valu only, 32 fma/visit", recs, idx, nblocks, visits, out, lds);
        run<1, 16>("smem, 32 fma/visit", recs, idx, nblocks, visits, out, lds);
        run<2, 16>("lds broadcast, 32 fma/visit", recs, idx, nblocks, visits, out, lds);
        run<3, 16>("smem asm prefetch, 32 fma/visit", recs, idx, nblocks, visits, out, lds);
        run<3, 4>("smem asm prefetch, 8 fma/visit", recs, idx, nblocks, visits, out, lds);
        run<1, 4>("smem, 8 fma/visit", recs, idx, nblocks, visits, out, lds);
        run<2, 4>("lds broadcast, 8 fma/visit", recs, idx, nblocks, visits, out, lds);
    }
    return 0;
}
