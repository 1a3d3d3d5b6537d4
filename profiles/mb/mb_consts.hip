// Micro-benchmark 2 (round 3): how should a wave get the per-triangle scan constants of its visits?
// Structure of the planned block-wave kernel: a ROUND = 64 candidates (lane = candidate gathers 96 B of its 128-B record with
// vector loads, as the cull test would), then the survivors (every other lane) are visited in lane order; a visit runs NV fp64
// FMAs on 12 wave-uniform doubles of the survivor's record.  Three ways to make them wave-uniform:
//   mode 1  s_load_dwordx16 + x8 from recs[tri] (L2-warm: the gather just touched the line), one visit ahead
//   mode 2  the survivor lanes write 96 B to LDS slots (16 slots per wave, sub-rounds), every lane reads the slot back (broadcast)
//   mode 3  v_readlane x 24 from the candidate lane's registers
//   mode 0  no constants at all (the FMAs use loop-invariant values): the floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
struct alignas(128) Rec { double d[16]; };
typedef const __attribute__((address_space(4))) double cdouble;
constexpr int SLOTS = 16;

template <int MODE, int NV, int WPB>
__global__ __launch_bounds__(64 * WPB) void k(const Rec* __restrict__ recs, const uint32_t* __restrict__ idx, int rounds, double* __restrict__ out, int lds_pad) {
    extern __shared__ double pad[];
    __shared__ __attribute__((aligned(16))) double slot[MODE == 2 ? WPB : 1][SLOTS][12];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t* my = idx + (size_t)(blockIdx.x * WPB + w) * rounds * 64;
    double acc0 = lane, acc1 = 1.0 + lane;
    if (lds_pad > 1 << 30) acc0 += pad[lane];
    for (int r = 0; r < rounds; ++r) {
        const uint32_t tri = my[r * 64 + lane];
        // ---- "cull": lane = candidate, 96 B of its record
        const double2* p = (const double2*)&recs[tri];
        double2 q[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) q[c] = p[c];
        double qs = 0; for (int c = 0; c < 6; ++c) qs += q[c].x + q[c].y;
        const bool surv = ((lane & 1) == 0) && (qs != 12345.0);
        unsigned long long todo = __ballot(surv);
        if (MODE == 0) {
            double c[12];
            for (int i = 0; i < 12; ++i) c[i] = 1.0 + 1e-9 * i;
            while (todo) {
                todo &= todo - 1;
#pragma unroll
                for (int o = 0; o < NV; ++o) { acc0 = __builtin_fma(acc0, c[o % 12], acc1); acc1 = __builtin_fma(acc1, c[(o + 5) % 12], acc0); }
                asm volatile("" : "+v"(acc0), "+v"(acc1));
            }
        } else if (MODE == 1) {
#if __HIP_DEVICE_COMPILE__
            double cur[12], nxt[12];
            if (todo) {
                int j = __builtin_ctzll(todo);
                uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)tri, j);
                cdouble* cp = (cdouble*)(recs + t);
                for (int i = 0; i < 12; ++i) cur[i] = cp[i];
            }
            while (todo) {
                todo &= todo - 1;
                {
                    const int j = todo ? __builtin_ctzll(todo) : 0;
                    const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)tri, j);
                    cdouble* cp = (cdouble*)(recs + t);
                    for (int i = 0; i < 12; ++i) nxt[i] = cp[i];
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int o = 0; o < NV; ++o) { acc0 = __builtin_fma(acc0, cur[o % 12], acc1); acc1 = __builtin_fma(acc1, cur[(o + 5) % 12], acc0); }
                __builtin_amdgcn_sched_barrier(0);
                for (int i = 0; i < 12; ++i) cur[i] = nxt[i];
            }
#endif
        } else if (MODE == 2) {
            const unsigned long long below = (1ull << lane) - 1ull;
            while (todo) {
                const uint32_t rank = __popcll(todo & below);
                const bool mine = ((todo >> lane) & 1ull) && rank < SLOTS;
                if (mine) {
                    double2* d = (double2*)slot[w][rank];
#pragma unroll
                    for (int c = 0; c < 6; ++c) d[c] = q[c];
                }
                const unsigned long long round = __ballot(mine);
                todo &= ~round;
                const int n = __popcll(round);
                __builtin_amdgcn_wave_barrier();
                for (int s = 0; s < n; ++s) {
                    const double2* d = (const double2*)slot[w][s];
                    double c[12];
#pragma unroll
                    for (int i = 0; i < 6; ++i) { const double2 t = d[i]; c[2 * i] = t.x; c[2 * i + 1] = t.y; }
#pragma unroll
                    for (int o = 0; o < NV; ++o) { acc0 = __builtin_fma(acc0, c[o % 12], acc1); acc1 = __builtin_fma(acc1, c[(o + 5) % 12], acc0); }
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            while (todo) {
                const int j = __builtin_ctzll(todo);
                todo &= todo - 1;
                double c[12];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    c[2 * i] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(q[i].x), j), __builtin_amdgcn_readlane(__double2loint(q[i].x), j));
                    c[2 * i + 1] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(q[i].y), j), __builtin_amdgcn_readlane(__double2loint(q[i].y), j));
                }
#pragma unroll
                for (int o = 0; o < NV; ++o) { acc0 = __builtin_fma(acc0, c[o % 12], acc1); acc1 = __builtin_fma(acc1, c[(o + 5) % 12], acc0); }
            }
        }
    }
    out[(size_t)blockIdx.x * 64 * WPB + threadIdx.x] = acc0 + acc1;
}

template <int MODE, int NV>
void run(const char* name, const Rec* recs, const uint32_t* idx, int nwaves, int rounds, double* out, int lds_bytes) {
    constexpr int WPB = 4;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int nblocks = nwaves / WPB;
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<MODE, NV, WPB>), dim3(nblocks), dim3(64 * WPB), lds_bytes, 0, recs, idx, rounds, out, lds_bytes);
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k<MODE, NV, WPB>), dim3(nblocks), dim3(64 * WPB), lds_bytes, 0, recs, idx, rounds, out, lds_bytes);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    const double nvis = (double)nwaves * rounds * 32;
    printf("%-28s %2d fma/visit  dyn lds %6d B/block: %8.3f ms  %6.1f M visits  %6.1f cycles/visit/SIMD\n", name, 2 * NV, lds_bytes, ms,
           nvis * 1e-6, ms * 1e-3 * 2.4e9 * 1024 / nvis);
}

int main() {
    const size_t nrec = 10u << 20;            // 1.34 GB of records
    const int rounds = 3, nwaves = 262144;    // 262144 waves x 3 rounds x 32 survivors = 25 M visits, 50 M candidates
    Rec* recs; CK(hipMalloc(&recs, nrec * sizeof(Rec)));
    {
        std::vector<double> h(16 * 4096);
        for (size_t i = 0; i < h.size(); ++i) h[i] = 1.0 + 1e-7 * (double)(i % 977);
        for (size_t o = 0; o < nrec; o += 4096) CK(hipMemcpy(recs + o, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    }
    std::vector<uint32_t> h((size_t)nwaves * rounds * 64);
    uint64_t st = 12345;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(st >> 33); };
    // the 16 waves of a "tile" (4 consecutive blocks) draw their candidates from the tile's pool of 1100 records, sorted ascending as a tile list is
    for (int t = 0; t < nwaves / 16; ++t) {
        std::vector<uint32_t> pool(1100);
        for (auto& p : pool) p = rnd() % nrec;
        for (int w = 0; w < 16; ++w)
            for (int r = 0; r < rounds * 64; ++r) h[((size_t)t * 16 + w) * rounds * 64 + r] = pool[rnd() % pool.size()];
    }
    uint32_t* idx; CK(hipMalloc(&idx, h.size() * 4)); CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    double* out; CK(hipMalloc(&out, (size_t)nwaves * 64 * 8));
    for (int lds : {0, 26000, 40000}) {       // 0: as many blocks per CU as registers allow; 26000: 6 blocks = 6 waves/SIMD; 40000: 4 blocks/CU = 4 waves/SIMD
        run<0, 10>("no constants", recs, idx, nwaves, rounds, out, lds);
        run<1, 10>("smem, one visit ahead", recs, idx, nwaves, rounds, out, lds);
        run<2, 10>("lds slots, broadcast", recs, idx, nwaves, rounds, out, lds);
        run<3, 10>("v_readlane x24", recs, idx, nwaves, rounds, out, lds);
        run<0, 16>("no constants", recs, idx, nwaves, rounds, out, lds);
        run<1, 16>("smem, one visit ahead", recs, idx, nwaves, rounds, out, lds);
        run<2, 16>("lds slots, broadcast", recs, idx, nwaves, rounds, out, lds);
        run<3, 16>("v_readlane x24", recs, idx, nwaves, rounds, out, lds);
    }
    return 0;
}
