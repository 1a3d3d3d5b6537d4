// kernels_raster.hip — the block rasterizer (gfx950): bounding-box scan, barycentric coverage test,
// fp64 z-test and the fragment shader of our_gl.cpp:147-199, one WAVEFRONT per 8x8-pixel block of a 32x32 screen tile.
//
// The block's z-buffer (fp64), colours and owners live in the wave's REGISTERS for the whole kernel (lane = pixel), so the
// compare-and-write of our_gl.cpp:165,191 needs no atomics and no memory at all: one wave walks the tile's triangle list in
// submission order and every lane owns a different pixel.  Depths and colours leave the chip once, at the end, as rows of the
// block; PHONG / EYE pixels are shaded once per visible pixel afterwards (k_shade).
//
// Arithmetic is the reference's, operation for operation, in fp64 with contraction off.  The three
// IEEE divisions per pixel of barycentric() (our_gl.cpp:85) matter because the `>= 0` coverage test
// and the written z depend on their rounding; for well-scaled triangles the same bits come from sign
// tests and FMA divisions by the per-triangle constant u.z ("exactness").
#include <hip/hip_runtime.h>
#include "trgl_device.h"
#include "launch.h"

namespace {

__device__ __forceinline__ double dmax(double a, double b) { return (a < b) ? b : a; }   // std::max
__device__ __forceinline__ double dmin(double a, double b) { return (b < a) ? b : a; }   // std::min
// v_max_f64 as one instruction (a NaN operand yields the other one): for the depth maxima, where a pixel holding NaN can
// never be written again (z < NaN is false), so leaving it out of a maximum keeps the maximum a valid bound
__device__ __forceinline__ double vmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double vmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// v_min3_f32 / v_max_f32 as single instructions (a NaN operand is skipped)
__device__ __forceinline__ float fmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float fmax2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// maximum over the 64 lanes, in every lane (NaN operands are skipped by v_max_f32)
__device__ __forceinline__ float wave_max_f32(float v) {
    float t;
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(t) : "v"(v)); v = t;
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=&v"(t) : "v"(v)); v = t;
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=&v"(t) : "v"(v)); v = t;
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=&v"(t) : "v"(v)); v = t;
    // every lane now holds the maximum of its row of 16: fold row 0 into 1 and 2 into 3, then rows 0-1 into 2-3; lane 63 has it all
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" : "=&v"(t) : "v"(v), "0"(v)); v = t;
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" : "=&v"(t) : "v"(v), "0"(v)); v = t;
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// double -> float rounded towards +inf / -inf (NaN stays NaN): bounds that stay bounds in single precision
__device__ __forceinline__ float f32_up(double d) {
    float f = (float)d;
    if ((double)f < d) { const uint32_t b = __float_as_uint(f); f = __uint_as_float((b >> 31) ? b - 1u : b + 1u); }
    return f;
}
__device__ __forceinline__ float f32_down(double d) {
    float f = (float)d;
    if ((double)f > d) { const uint32_t b = __float_as_uint(f); f = __uint_as_float((b >> 31) ? b + 1u : b - 1u); }
    return f;
}
__device__ __forceinline__ int iclamp(int v, int lo, int hi) { return (v < lo) ? lo : (hi < v) ? hi : v; }
__device__ __forceinline__ int x86_cvttsd2si(double d) {
    if (!(d > -2147483649.0 && d < 2147483648.0)) return INT_MIN;
    return (int)d;
}
__device__ __forceinline__ double dot3(const double* a, const double* b) {
    double sum = 0; sum += a[0] * b[0]; sum += a[1] * b[1]; sum += a[2] * b[2]; return sum;   // geometry.h:122-127
}
__device__ __forceinline__ void normalized3(const double* v, double* out) {                   // geometry.h:136-140
    double length = sqrt(dot3(v, v));
    if (length == 0) { out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; return; }
    out[0] = v[0] / length; out[1] = v[1] / length; out[2] = v[2] / length;
}
__device__ __forceinline__ unsigned long long zkey(double d) {       // order-preserving u64 key of a double
    unsigned long long b = (unsigned long long)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// ---- samplers: model.cpp:415-459 + TGAImage::get tgaimage.cpp:24-30 ----------------------------
struct Color { uint32_t bgra; int bytespp; };   // TGAColor (tgaimage.h:29-31), bgra[0] in the low byte

// (TX: pointer to DevTexture in the generic or in the constant address space - k_shade reads descriptors through scalar loads)
template <class TX>
__device__ __forceinline__ TX tex_slot(TX tex, int slot) {
    if (slot < 0 || slot >= TRGL_MAX_TEXTURES) return nullptr;
    if (!tex[slot].data || tex[slot].w <= 0) return nullptr;
    return &tex[slot];
}
// TGAImage::get at the clamped texel (model.cpp:420-425 etc.): ONE unaligned 4-byte load per texel (the device copy
// of every texture is padded by 4 bytes), masked to bpp bytes = TGAColor(p, bpp) with the rest 0 (tgaimage.h:46-50).
template <class TX>
__device__ __forceinline__ uint32_t tex_fetch_raw(TX t, const double* uv) {
    int x = iclamp(x86_cvttsd2si(uv[0] * t->w), 0, t->w - 1);
    int y = iclamp(x86_cvttsd2si(uv[1] * t->h), 0, t->h - 1);
    const uint8_t* p = t->data + ((size_t)x + (size_t)y * t->w) * t->bpp;
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
template <class TX>
__device__ __forceinline__ uint32_t tex_mask(TX t) { return t->bpp >= 4 ? 0xffffffffu : ((1u << (8 * t->bpp)) - 1u); }
template <class TX>
__device__ __forceinline__ Color tex_fetch(TX t, const double* uv) {
    return Color{ tex_fetch_raw(t, uv) & tex_mask(t), t->bpp };
}
__device__ __forceinline__ void interp(const double* v0, const double* v1, const double* v2, const double* b, int n, double* out) {
    for (int i = 0; i < n; ++i) out[i] = (v0[i] * b[0] + v1[i] * b[1]) + v2[i] * b[2];       // main.cpp:94-104
}
__device__ __forceinline__ double spec_pow(double x, double y) {
    // std::pow(x, 1.0) == x exactly; PhongShader's exponent is always 1.0
    return (y == 1.0) ? x : pow(x, y);
}

// PhongShader::fragment — main.cpp:92-170
template <class UN, class TX>
__device__ __forceinline__ Color frag_phong(const UN& u, TX tx, const double* vary, const double* b) {
    const double* uvv = vary; const double* pos = vary + 6; const double* nrm = vary + 15;
    double position_eye[3], geometry_normal[3], uv[2];
    interp(pos, pos + 3, pos + 6, b, 3, position_eye);
    interp(nrm, nrm + 3, nrm + 6, b, 3, geometry_normal);
    interp(uvv, uvv + 2, uvv + 4, b, 2, uv);

    // the three maps are independent: issue all texel loads first, consume afterwards (the slots are per draw,
    // so these branches are wave-uniform)
    const TX td = tex_slot(tx, u.tex_diffuse);
    const TX ts = tex_slot(tx, u.tex_specular);
    const TX tn = tex_slot(tx, u.tex_normal);
    uint32_t raw_d = 0, raw_s = 0, raw_n = 0;
    if (td) raw_d = tex_fetch_raw(td, uv);
    if (ts) raw_s = tex_fetch_raw(ts, uv);
    if (tn) raw_n = tex_fetch_raw(tn, uv);
    Color base = td ? Color{ raw_d & tex_mask(td), td->bpp } : Color{ 0xffffffffu, 4 };       // model.cpp:415-426
    float specf = ts ? (float)(int)(raw_s & 0xff) / 255.0f : 1.0f;                            // model.cpp:447-459
    double specular_power = dmax(1.0, (double)specf);

    int bsum = (int)(base.bgra & 0xff) + (int)((base.bgra >> 8) & 0xff) + (int)((base.bgra >> 16) & 0xff);
    double brightness = bsum / (3.0 * 255.0);
    bool is_eye_pixel = (brightness >= 0.85) && (specular_power <= 5.0);

    double nmv[3] = { 0, 0, 1 };                                                              // model.cpp:428-445
    if (tn) {
        const uint32_t c = raw_n & tex_mask(tn);
        double n[3];
        n[0] = (double)((c >> 16) & 0xff) / 255.0 * 2.0 - 1.0;
        n[1] = (double)((c >> 8) & 0xff) / 255.0 * 2.0 - 1.0;
        n[2] = (double)(c & 0xff) / 255.0 * 2.0 - 1.0;
        normalized3(n, nmv);
    }
    double nme[3];                                                                            // main.cpp:116-119
    for (int r = 0; r < 3; ++r) {
        const auto* m = u.model_view + 4 * r;
        double sum = 0; sum += m[0] * nmv[0]; sum += m[1] * nmv[1]; sum += m[2] * nmv[2]; sum += m[3] * 0.0;
        nme[r] = sum;
    }
    double N[3];
    if (is_eye_pixel) { N[0] = geometry_normal[0]; N[1] = geometry_normal[1]; N[2] = geometry_normal[2]; }
    else {
        double s = u.normal_map_strength, mix[3];
        for (int i = 0; i < 3; ++i) mix[i] = geometry_normal[i] * (1.0 - s) + nme[i] * s;
        normalized3(mix, N);
    }
    double negp[3], V[3];
    for (int i = 0; i < 3; ++i) negp[i] = position_eye[i] * -1.0;
    normalized3(negp, V);

    const double Lk[3] = { u.key_light_dir_eye[0], u.key_light_dir_eye[1], u.key_light_dir_eye[2] };
    double key_diffuse = dmax(0.0, dot3(N, Lk)) * 1.0;
    double k2 = 2.0 * dot3(N, Lk), rr[3], R[3];
    for (int i = 0; i < 3; ++i) rr[i] = N[i] * k2 - Lk[i];
    normalized3(rr, R);
    double rvd = dmax(0.0, dot3(R, V));
    double key_specular = (rvd > 0.0 ? spec_pow(rvd, specular_power) : 0.0) * 1.0;
    const double Lf[3] = { u.fill_light_dir_eye[0], u.fill_light_dir_eye[1], u.fill_light_dir_eye[2] };
    double fill_diffuse = dmax(0.0, dot3(N, Lf)) * 0.35;
    const double Lr[3] = { u.rim_light_dir_eye[0], u.rim_light_dir_eye[1], u.rim_light_dir_eye[2] };
    double rim_diffuse = dmax(0.0, dot3(N, Lr)) * 0.6;
    double total_diffuse = key_diffuse + fill_diffuse + rim_diffuse;
    double total_specular = key_specular;
    double ambient = 0.10;

    uint32_t out = base.bgra & 0xff000000u;
    for (int ch = 0; ch < 3; ++ch) {
        double channel_value = (double)((base.bgra >> (8 * ch)) & 0xff);
        double final_value = channel_value * (ambient + total_diffuse) + 255.0 * (0.35 * total_specular);
        out |= (uint32_t)(unsigned char)dmin(255.0, final_value) << (8 * ch);
    }
    return Color{ out, base.bytespp };
}

// EyeShader::fragment — main.cpp:220-261
template <class UN, class TX>
__device__ __forceinline__ Color frag_eye(const UN& u, TX tx, const double* vary, const double* b) {
    const double* uvv = vary; const double* pos = vary + 6; const double* nrm = vary + 15;
    double position_eye[3], ni[3], N[3], uv[2];
    interp(pos, pos + 3, pos + 6, b, 3, position_eye);
    interp(nrm, nrm + 3, nrm + 6, b, 3, ni);
    normalized3(ni, N);
    interp(uvv, uvv + 2, uvv + 4, b, 2, uv);

    const TX td = tex_slot(tx, u.tex_diffuse);
    const TX ts = tex_slot(tx, u.tex_specular);
    uint32_t raw_d = 0, raw_s = 0;
    if (td) raw_d = tex_fetch_raw(td, uv);
    if (ts) raw_s = tex_fetch_raw(ts, uv);
    Color base = td ? Color{ raw_d & tex_mask(td), td->bpp } : Color{ 0xffffffffu, 4 };
    double negp[3], V[3];
    for (int i = 0; i < 3; ++i) negp[i] = position_eye[i] * -1.0;
    normalized3(negp, V);

    const double Lk[3] = { u.key_light_dir_eye[0], u.key_light_dir_eye[1], u.key_light_dir_eye[2] };
    double key_diffuse = dmax(0.0, dot3(N, Lk)) * 1.0;
    const double Lr[3] = { u.rim_light_dir_eye[0], u.rim_light_dir_eye[1], u.rim_light_dir_eye[2] };
    double rim_diffuse = dmax(0.0, dot3(N, Lr)) * 0.6;
    double total_diffuse = key_diffuse + rim_diffuse;

    float specf = ts ? (float)(int)(raw_s & 0xff) / 255.0f : 1.0f;
    double specular_power = dmax(1.0, (double)specf) * 8.0;
    double k2 = 2.0 * dot3(N, Lk), rr[3], R[3];
    for (int i = 0; i < 3; ++i) rr[i] = N[i] * k2 - Lk[i];
    normalized3(rr, R);
    double rvd = dmax(0.0, dot3(R, V));
    double specular = (rvd > 0.0 ? spec_pow(rvd, specular_power) : 0.0);

    uint32_t out = base.bgra & 0xff000000u;
    for (int ch = 0; ch < 3; ++ch) {
        double channel_value = (double)((base.bgra >> (8 * ch)) & 0xff);
        double final_value = channel_value * (0.1 + total_diffuse) + 255.0 * (1.5 * specular);
        out |= (uint32_t)(unsigned char)dmin(255.0, final_value) << (8 * ch);
    }
    return Color{ out, base.bytespp };
}

// GOURAUD: base * (float)(i0*b0 + i1*b1 + i2*b2), TGAColor::operator*(float) tgaimage.h:55-62
__device__ __forceinline__ uint32_t frag_gouraud(uint32_t base, const double* vary, const double* b) {
    double id = (vary[0] * b[0] + vary[1] * b[1]) + vary[2] * b[2];
    float intensity = (float)id;
    if (intensity < 0.f) intensity = 0.f;
    if (intensity > 1.f) intensity = 1.f;
    uint32_t out = 0;
    for (int i = 0; i < 4; ++i) {
        float c = (float)(int)((base >> (8 * i)) & 0xff) * intensity;
        out |= (uint32_t)(uint8_t)c << (8 * i);
    }
    return out;
}


// CHECKER: a FLAT colour that discards on a predicate of the perspective-correct barycentrics - the device kind that exercises
// `if (discard) continue;` (our_gl.cpp:187-188): cell parities of bar[0] and bar[1] on a cells x cells grid differ -> discard.
__device__ __forceinline__ bool frag_checker_discards(int cells, const double* b) {
    const int a = x86_cvttsd2si(b[0] * (double)cells), c = x86_cvttsd2si(b[1] * (double)cells);
    return ((a ^ c) & 1) != 0;
}

// a / uz, correctly rounded, for the per-triangle constant uz with ruz = RN(1/uz) (Markstein):
// q0 = RN(a*ruz) is within 2 ulp of a/uz; one FMA residual step makes q1 faithful (error < 1 ulp),
// and for a faithful q1 the second step q1 + (a - uz*q1)*ruz rounds to RN(a/uz) exactly.
// Valid when nothing over/underflows: only used for "well scaled" triangles (see k_setup).
__device__ __forceinline__ double div_by_uz(double a, double uz, double ruz) {
    const double q0 = a * ruz;
    const double e0 = __builtin_fma(-q0, uz, a);
    const double q1 = __builtin_fma(e0, ruz, q0);
    const double e1 = __builtin_fma(-q1, uz, a);
    return __builtin_fma(e1, ruz, q1);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_raster: ONE WAVEFRONT PER 8x8-PIXEL BLOCK, the block's depths, colours (and owners) in REGISTERS.
//
// A workgroup is four waves = one row of four blocks of a 32x32 tile; all sixteen block waves of a tile walk the tile's
// triangle list (submission order, from the binning) on their own, without barriers.  Lane l of a wave owns pixel
// (l & 7, l >> 3) of the block for the whole kernel: its depth is a register pair, the compare-and-write of our_gl.cpp:165,191
// is two register moves, nothing about a pixel ever goes through LDS, and every pixel leaves the chip exactly once at the end
// (depth 8 B, colour bpp B, written as whole rows of the block).
//
// Per wave:
//  1. candidates   The pairs of the tile carry a 4x4 mask of the tile's blocks that the triangle's clamped bbox reaches
//                  (k_expand); the wave takes the entries with its bit, 64 list entries per step, and compacts their triangle
//                  ids into a small ring in LDS (order preserved).
//  2. cull         Whenever 64 candidates are waiting (or the list ends): lane = candidate.  The lane reads 96 B of its
//                  triangle's record and drops the triangle when one of the three edge functions excludes every pixel of
//                  (bbox n block) or when its depth plane cannot get below the block's largest stored depth (both tests in fp64
//                  with margins 2^10 times the rounding they cover; "edge" and "depth" below).  The gather of a round is requested
//                  one round ahead: between the test of the round before it and the visits of that round's survivors.
//  3. visits       The survivors (compacted into the low lanes), in list order.  The triangle's constants arrive as wave-uniform
//                  values through the scalar cache (two s_load_dwordx16 = the whole record, requested one visit ahead), so the
//                  vector instructions take them as scalar operands and no vector register holds a per-triangle constant.
//                  Per visit, one pixel per lane: depth plane against the lane's stored depth, barycentric() + sign coverage test
//                  (no bbox test: k_setup sends the slivers that would need one down the literal path).  That is where a visit
//                  ENDS: the covered lanes only note (u.x, u.y, triangle) in registers ...
//  4. resolve      ... and the three divisions of our_gl.cpp:85, the depth of :156-158, the z-test and the fragment run later
//                  for ALL noted lanes at once, each lane with its own triangle (its u.z, 1/u.z, depths and colour were noted too:
//                  no memory access): when a visit covers a lane that still holds a note (so that every pixel sees its
//                  fragments in submission order) and at the end of the list.  Small triangles cover a few lanes each;
//                  resolving several of them together is what fills the lanes of the most expensive part of the pixel loop.
//
// The scalar unit issues one instruction per cycle for the four SIMDs of a CU and was as busy as the vector units in the first
// version of this kernel: the visit loop is written to need few scalar instructions (k_raster).
//
// Exactness: as before, operation for operation in fp64 with contraction off; sign coverage + FMA division by u.z for
// well-scaled triangles ("exactness"), the literal divisions for the others (`dl` bit 31).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int KIND_ANY = TRGL_NUM_SHADERS;          // per-fragment switch on the draw's kind (mixed flushes)
constexpr int RING = 512;                           // candidate ring of a wave (triangle ids): a power of two >= 63 left over + 256 of a step

#ifdef TRGL_DEBUG_COUNTERS
#define TRGL_DBG(i, n) do { S.dbg[i] += (unsigned long long)(n); } while (0)
// phase clocks of the diagnostic build: the cycles since the last stamp go to counter i
#ifdef TRGL_DEBUG_NOSTAMP      // (counters 10-13 then describe the first list entry that is no triangle, profiles/one_case.py)
#define TRGL_STAMP(i) ((void)0)
#else
#define TRGL_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); S.dbg[i] += t_ - S.t_prev; S.t_prev = t_; } while (0)
#endif
#else
#define TRGL_DBG(i, n) ((void)0)
#define TRGL_STAMP(i) ((void)0)
#endif

typedef const __attribute__((address_space(4))) TriRec CRec;      // records through the scalar cache
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// The scan constants of one triangle, wave-uniform (SGPRs): its whole record, two scalar loads of 64 B.
struct TriScan {
    double ax, ay, s0x, s0y, s1x, s1y, c0, uz, g1, g2;
    uint32_t bx, by;      // as stored: bx0 | by0 << 16, bx1 | by1 << 16
    uint32_t color, dl;
    double ruz, z0, z1, z2;       // ride along for the lanes that note a fragment (see BlockState)
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    u32x16 lo, hi;                // the 32 registers as requested: all of them stay allocated until scan_wait()
};
__device__ __forceinline__ TriScan load_scan(const TriRec* __restrict__ recs, uint32_t tri) {
    TriScan T;
#ifdef TRGL_EXP_SAMEREC
    tri &= 63u;          // timing experiment only (wrong frames): every visit reads one of 64 records = scalar-cache hits
#endif
#if __HIP_DEVICE_COMPILE__
    // (a flush holds at most TRGL_FLUSH_MAX_TRIS = 2^25 records: the byte offset fits the 32-bit scalar offset of s_load)
    // The record as TWO requests of 16 dwords, written out: from C++ the second half becomes three (8 + 4 + 2 dwords around the bbox,
    // which the fast visit does not use).  The compiler does not know that these registers arrive later: every use is preceded by
    // scan_wait() below, and nothing else may touch them in between (the two sets A and B of the visit loop are never copied).
    TriScan::u32x16 g, h;
    asm volatile("s_load_dwordx16 %0, %2, %3 offset:0x0\n\ts_load_dwordx16 %1, %2, %3 offset:0x40" : "=&s"(g), "=&s"(h) : "s"(recs), "s"(tri << 7));
    T.lo = g; T.hi = h;
#ifdef TRGL_DEBUG_COUNTERS
    // (the diagnostic build keeps so many counters in scalar registers that the compiler spills a set while its request is still out -
    // tools/check_scan_regs.py refuses that; here the request is waited for at once, which costs the visits of THIS build their overlap)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(T.lo), "+s"(T.hi) :: "memory");
    g = T.lo; h = T.hi;
#endif
    auto dg = [&](int i) { return __builtin_bit_cast(double, ((unsigned long long)g[i + 1] << 32) | g[i]); };
    auto dh = [&](int i) { return __builtin_bit_cast(double, ((unsigned long long)h[i + 1] << 32) | h[i]); };
    T.ax = dg(0); T.ay = dg(2); T.s0x = dg(4); T.s0y = dg(6); T.s1x = dg(8); T.s1y = dg(10); T.c0 = dg(12); T.uz = dg(14);
    T.g1 = dh(0); T.g2 = dh(2); T.ruz = dh(4); T.z0 = dh(6); T.z1 = dh(8); T.z2 = dh(10);
    T.bx = h[12]; T.by = h[13]; T.color = h[14]; T.dl = h[15];
#endif
    return T;
}
// (the registers of a set are inputs of the wait: a field that the kernel variant never reads - the bbox, colour or `dl` - would otherwise
// be dead from the start, and its register handed to something else while the load that writes it is still in flight)
__device__ __forceinline__ void scan_wait(const TriScan& T) { asm volatile("s_waitcnt lgkmcnt(0)" :: "s"(T.lo), "s"(T.hi) : "memory"); }

// Per-wave state: the block's pixels (one per lane) and the running statistics of our_gl.cpp:194-198.
struct BlockState {
    double z;               // stored depth of the lane's pixel (-inf: a pixel this wave does not own - never written, never stored)
    uint32_t color;         // its colour, b | g << 8 | r << 16 | a << 24
    uint32_t id;            // PHONG / EYE flushes: draw << 24 | triangle of the fragment that owns the pixel, ~0u = none (k_shade)
    double pxc, pyc;        // pixel centre (x + 0.5, y + 0.5), our_gl.cpp:149
    // deferred fragments: lanes of `pend` hold (u.x, u.y) of barycentric(), their triangle and - copied from the wave-uniform
    // constants of the visit, so that resolving them needs no memory access at all - its u.z, 1/u.z and the three vertex depths
    double pux, puy; uint32_t ptri;
    double puz, pruz, pz0, pz1, pz2;
    uint32_t pcd;           // ... and its colour (FLAT flushes) or `dl` (all others)
    uint32_t frags; double zmax;   // z range of the fragments the lane wrote: the largest one is kept; the smallest IS the pixel's
                                   // final depth (every write lowers it), read off at block-out
    us2 xy;                 // x | y << 16 as two 16-bit words (the bbox test of a visit is packed 16-bit arithmetic)
#ifdef TRGL_DEBUG_COUNTERS
    unsigned long long dbg[16];    // work counters of the diagnostic build (profiles/raster_counters.py); [8] is summed over the lanes
    unsigned long long t_prev;
#endif
};

// Resolve the deferred fragments: our_gl.cpp:85 (the three quotients), :156-165 (depth, finite check, strict z-test), :168-192
// (perspective-correct barycentrics, fragment, writes) and :194-198 (counters), each lane for its own triangle.
template <int KIND, bool ALLWS, bool DEFERRED>
__device__ __forceinline__ void resolve(BlockState& S, unsigned long long pend, const TriRec* __restrict__ recs, const TriW* __restrict__ recs_w,
                                        const DrawDesc* __restrict__ draws, DevStats* __restrict__ stats, bool zero_locked) {
    TRGL_DBG(6, 1); TRGL_DBG(7, __popcll(pend));
#ifdef TRGL_DEBUG_COUNTERS
    const unsigned long long t_res = __builtin_amdgcn_s_memtime();
#endif
    // PLAIN: every fragment of the flush is FLAT, PHONG or EYE: its colour / owner id depends on the triangle only and rides with the
    // deferred fragment (S.pcd), so resolving touches no memory at all.
    constexpr bool PLAIN = KIND == TRGL_SHADER_FLAT || KIND == TRGL_SHADER_PHONG || KIND == TRGL_SHADER_EYE;
    if (__builtin_amdgcn_inverse_ballot_w64(pend)) {
        const double uz = S.puz, ruz = S.pruz, z0 = S.pz0, z1 = S.pz1, z2 = S.pz2;
        const double ux = S.pux, uy = S.puy, us = ux + uy;
        double b0, b1, b2;
        if (ALLWS || ruz != 0.0) {
            b0 = 1.0 - div_by_uz(us, uz, ruz);                              // :85 through the reciprocal ("exactness")
            b1 = div_by_uz(uy, uz, ruz);
            b2 = div_by_uz(ux, uz, ruz);
        } else {
            b0 = 1.0 - us / uz;                                             // :85, as written
            b1 = uy / uz;
            b2 = ux / uz;
        }
        const double z = b0 * z0 + b1 * z1 + b2 * z2;                       // :156-158
        if (__builtin_isfinite(z) && (z < S.z)) {                           // :160, :165
            uint32_t color = S.pcd, id = 0xffffffffu;
            bool discard = false;
            if (KIND == TRGL_SHADER_PHONG || KIND == TRGL_SHADER_EYE) id = TRGL_DL_ID(S.pcd);
            if (!PLAIN) {
                const uint32_t dl = S.pcd;
                color = recs[S.ptri].color;
                const DrawDesc* d = draws + TRGL_DL_DRAW(dl);
                const int kind = KIND == KIND_ANY ? d->kind : KIND;
                if (kind == TRGL_SHADER_PHONG || kind == TRGL_SHADER_EYE) {
                    // not shaded here: the pixel remembers which triangle owns it and k_shade runs the fragment shader once per
                    // visible pixel (the two shaders have no side effects and never discard, so the image is the same as shading
                    // every z-pass in order, and the counters do not depend on colours)
                    id = TRGL_DL_ID(dl);
                } else if (kind != TRGL_SHADER_FLAT) {
                    const TriW w = recs_w[S.ptri];
                    double pc[3];
                    const double denom = b0 * w.iw0 + b1 * w.iw1 + b2 * w.iw2;                    // :172-174
                    if (fabs(denom) < 1e-15) { pc[0] = b0; pc[1] = b1; pc[2] = b2; }              // :177-185
                    else { pc[0] = (b0 * w.iw0) / denom; pc[1] = (b1 * w.iw1) / denom; pc[2] = (b2 * w.iw2) / denom; }
                    if (kind == TRGL_SHADER_GOURAUD) {
                        const double* vary = d->vary + (size_t)TRGL_DL_LOCAL(dl) * 3;
                        const double vv[3] = { vary[0], vary[1], vary[2] };
                        color = frag_gouraud(color, vv, pc);
                    } else {                                                                      // TRGL_SHADER_CHECKER
                        discard = frag_checker_discards(d->u.reserved, pc);                       // our_gl.cpp:187-188
                    }
                }
            }
            if (!discard) {
                // :191-198, committed IN PLACE (asm operands tied to the state registers): written through ordinary assignments the
                // compiler gives every state variable a second register inside the conditional resolve and copies all of them back at
                // its end - nine moves per covered visit.  v_max gives std::max up to the sign of a zero, which k_fold_stats settles
                // from the first-zero keys below (as it does for the minimum, taken from the final depths at block-out).
                asm volatile(
                    "v_mov_b64 %[sz], %[z]\n\t"                                                // :191
                    "v_add_u32 %[fr], 1, %[fr]\n\t"                                            // :194
                    "v_max_f64 %[zmax], %[zmax], %[z]"                                           // :198
                    : [sz] "+v"(S.z), [fr] "+v"(S.frags), [zmax] "+v"(S.zmax)
                    : [z] "v"(z));
                if (KIND == TRGL_SHADER_PHONG || KIND == TRGL_SHADER_EYE) {
                    asm volatile("v_mov_b32 %0, %1" : "+v"(S.id) : "v"(id));                    // (the colour comes from k_shade)
                } else {
                    const uint32_t ncol = (!DEFERRED || id == 0xffffffffu) ? color : S.color;   // :192 (tgaimage.cpp:32-39 at block-out)
                    asm volatile("v_mov_b32 %0, %1" : "+v"(S.color) : "v"(ncol));
                    if (DEFERRED) asm volatile("v_mov_b32 %0, %1" : "+v"(S.id) : "v"(id));
                }
                TRGL_DBG(8, 1);
                // std::min / std::max keep the first of equal values and +0.0 == -0.0: when the z range ends in a zero its sign is that
                // of the first zero written in the reference's order (triangle, x, y), see DevStats
                if (z == 0.0 && !zero_locked) {
                    const uint32_t xy = __builtin_bit_cast(uint32_t, S.xy);
                    unsigned long long order = ((unsigned long long)S.ptri << 32) | ((unsigned long long)(xy & 0xffffu) << 16) | (unsigned long long)(xy >> 16);
                    atomicMin(__builtin_signbit(z) ? &stats->zero_neg_key : &stats->zero_pos_key, order);
                }
            }
        }
    }
#if defined(TRGL_DEBUG_COUNTERS) && !defined(TRGL_DEBUG_NOSTAMP)
    S.dbg[14] += __builtin_amdgcn_s_memtime() - t_res;
#endif
}

// Rows of a cleared tile without triangles: the clear values, row-contiguous (this is the whole kernel on a clear-only frame, the
// "framebuffer + z write-out" figure of BASELINE.json).  Wave w of the workgroup stores rows [py0 + 8 w, py0 + 8 w + 7].
__device__ __forceinline__ void clear_rows(const FrameParams& fp, int lane, int px0, int y0, int y1) {
    const int xa1 = min(px0 + TRGL_TILE - 1, fp.W - 1);
    const bool full_x = (px0 + TRGL_TILE - 1) <= xa1;
    // z: 16 B per lane, 4 rows per store instruction
    if (full_x && (fp.W & 1) == 0) {
        for (int r4 = 0; r4 < 8; r4 += 4) {
            const int x = px0 + ((lane & 15) << 1), y = y0 + r4 + (lane >> 4);
            if (y <= y1) {
                typedef double nt_d2 __attribute__((ext_vector_type(2)));
                nt_d2 nv = { fp.clear_z, fp.clear_z };
                __builtin_nontemporal_store(nv, reinterpret_cast<nt_d2*>(&fp.zb[(size_t)x + (size_t)y * fp.W]));
            }
        }
    } else {
        for (int r2 = 0; r2 < 8; r2 += 2) {
            const int x = px0 + (lane & 31), y = y0 + r2 + (lane >> 5);
            if (x <= xa1 && y <= y1) fp.zb[(size_t)x + (size_t)y * fp.W] = fp.clear_z;
        }
    }
    // colour: 4 pixels per lane (12 B for RGB, 16 B for RGBA), 8 rows per store instruction
    if (full_x && (fp.W & 3) == 0 && (fp.bpp == 3 || fp.bpp == 4)) {
        const int x = px0 + ((lane & 7) << 2), y = y0 + (lane >> 3);
        if (y <= y1) {
            const uint32_t c = fp.clear_color;
            const size_t idx = (size_t)x + (size_t)y * fp.W;
            if (fp.bpp == 4) {
                *reinterpret_cast<uint4*>(fp.fb + idx * 4) = make_uint4(c, c, c, c);
            } else {
                uint32_t* dst = reinterpret_cast<uint32_t*>(fp.fb + idx * 3);
                dst[0] = (c & 0xffffffu) | (c << 24); dst[1] = ((c >> 8) & 0xffffu) | (c << 16); dst[2] = ((c >> 16) & 0xffu) | (c << 8);
            }
        }
    } else {
        for (int r2 = 0; r2 < 8; r2 += 2) {
            const int x = px0 + (lane & 31), y = y0 + r2 + (lane >> 5);
            if (x <= xa1 && y <= y1) {
                const uint32_t c = fp.clear_color;
                uint8_t* dst = fp.fb + ((size_t)x + (size_t)y * fp.W) * fp.bpp;
                for (int i = 0; i < fp.bpp; ++i) dst[i] = (uint8_t)(c >> (8 * i));
            }
        }
    }
    if (fp.idbuf) {
        for (int r2 = 0; r2 < 8; r2 += 2) {
            const int x = px0 + (lane & 31), y = y0 + r2 + (lane >> 5);
            if (x <= xa1 && y <= y1) fp.idbuf[(size_t)x + (size_t)y * fp.W] = 0xffffffffu;
        }
    }
}

// Work items (k_make_items): one per workgroup.
//   bits 0-23 tile, bits 24-25 row of blocks inside the tile, bit 31: the tile has no triangles and is only cleared
#define TRGL_ITEM_CLEAR 0x80000000u
#ifndef TRGL_RASTER_WAVES
#define TRGL_RASTER_WAVES 5        // waves per SIMD the register allocation of k_raster aims at (96 vector registers; 6 waves = 80 registers spill 8 dwords per list step: 2 % slower)
#endif

// BPP: the framebuffer's bytes per pixel when the kernel is compiled for one (3 or 4, FLAT only), 0 = read from FrameParams
// ALLWS: the flush holds no triangle that needs the literal path (k_setup counted them): the kernel is compiled without it
// DEFERRED: the flush has PHONG / EYE draws (visibility buffer + k_shade)
template <int KIND, int BPP = 0, bool ALLWS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TRGL_RASTER_WAVES, 8))) void k_raster(FrameParams fp, const TriRec* __restrict__ recs, const TriW* __restrict__ recs_w,
                                                const uint32_t* __restrict__ vals, const uint16_t* __restrict__ bmask,
                                                const uint32_t* __restrict__ tile_start,
                                                const uint32_t* __restrict__ tile_end,
                                                const DrawDesc* __restrict__ draws, DevStats* __restrict__ stats,
                                                const uint4* __restrict__ items, const uint32_t* __restrict__ n_items,
                                                unsigned long long* __restrict__ item_stats) {
    constexpr bool DEFERRED = KIND == TRGL_SHADER_PHONG || KIND == TRGL_SHADER_EYE || KIND == KIND_ANY;
    __shared__ uint32_t s_ring[4][RING];
    __shared__ uint32_t s_out[4][64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): workgroup b runs on the XCD of b mod 8.  Give every
    // XCD one contiguous eighth of the item list, so that the four block rows of a tile - and its neighbours, which share
    // triangles with it - read their list and records through the same L2.  (Placement is an observed property, used for speed only.)
    const uint32_t G = n_items[0];
    const uint32_t per = (G + 7u) >> 3, xj = blockIdx.x >> 3;
    const uint32_t g = (blockIdx.x & 7u) * per + xj;
    if (xj >= per || g >= G) return;                      // (block-uniform, like the clear-only exit below: the one barrier at the end is safe)
    const uint4 item4 = items[g];                         // work item, its tile's slice [beg, end) of the sorted pair list
    const uint32_t item = item4.x;
    const int t = (int)(item & 0xffffffu);
    const int tile_y = t / fp.tiles_x, tile_x = t - tile_y * fp.tiles_x;
    const int px0 = tile_x << TRGL_TILE_LOG2, py0 = tile_y << TRGL_TILE_LOG2;
    unsigned long long* my_stats = item_stats + (size_t)g * 4;
    if (item & TRGL_ITEM_CLEAR) {                         // cleared and empty: store the clear values, nothing else
        // rows [py0 + 8 w, py0 + 8 w + 7] of the tile for wave w
        const int ya = max(py0 + 8 * w, fp.strip_y0), yb = min(min(py0 + 8 * w + 7, fp.H - 1), fp.strip_y1 - 1);
        if (ya <= yb) clear_rows(fp, lane, px0, ya, yb);
        if (threadIdx.x == 0) {
            ulonglong2* dst = reinterpret_cast<ulonglong2*>(my_stats);
            dst[0] = make_ulonglong2(0ull, ~0ull); dst[1] = make_ulonglong2(0ull, 0ull);
        }
        return;
    }
    const int brow = (int)((item >> 24) & 3u);
    const int kblk = 4 * brow + w;                        // this wave's block of the tile: bit 4 cy + cx of the pair masks
    const int X0 = px0 + 8 * w, Y0 = py0 + 8 * brow;

    BlockState S;
    const int lx = X0 + (lane & 7), ly = Y0 + (lane >> 3);          // the lane's pixel
    S.xy = __builtin_bit_cast(us2, (uint32_t)lx | ((uint32_t)ly << 16));
    const bool owned = lx < fp.W && ly < fp.H && ly >= fp.strip_y0 && ly < fp.strip_y1;
    const size_t pix = (size_t)lx + (size_t)ly * fp.W;
    S.pxc = (double)lx + 0.5; S.pyc = (double)ly + 0.5;
    // Pixels of the block that this context does not own (rows outside the strip, beyond the image) hold -inf: no fragment
    // passes there, and they are not stored.
    S.z = -__builtin_inf();
    if (owned) S.z = fp.init_from_clear ? fp.clear_z : fp.zb[pix];
    S.color = fp.clear_color;
    S.id = 0xffffffffu;
    S.pux = 0.0; S.puy = 0.0; S.ptri = 0;
    S.puz = -1.0; S.pruz = -1.0; S.pz0 = 0.0; S.pz1 = 0.0; S.pz2 = 0.0; S.pcd = 0;
    S.frags = 0; S.zmax = -__builtin_inf();
    const bool zero_locked = stats->zero_locked != 0;
#ifdef TRGL_DEBUG_COUNTERS
    for (int k = 0; k < 16; ++k) S.dbg[k] = 0;
    S.t_prev = __builtin_amdgcn_s_memtime();
#endif
    unsigned long long pend = 0;                          // lanes that hold a deferred fragment (wave-uniform)

    const uint32_t beg = item4.y, end = item4.z;
    uint32_t* ring = s_ring[w];
    uint32_t head = 0, cnt = 0;                           // ring: `cnt` candidates wait from position `head` on
    // The list, 256 entries per step: lane l takes entries 4 l .. 4 l + 3 of the step (one 16-byte load of triangle ids, one 8-byte
    // load of block masks; steps start at a multiple of 4 entries, the pair buffers hold a multiple of 4).  The entries of the next
    // step are requested before the candidates of this one are processed.  Loads without a branch around them (clamped index, masks
    // cleared outside [beg, end)), so that the wait for THIS step's entries can leave the next step's loads in flight.
    // the round whose gather is in flight: its candidates (lane = candidate), how many, and chunks 0-4 and 7 of their records
    uint32_t pn = 0, ptri = 0;
    double2 pq0 = make_double2(0, 0), pq1 = pq0, pq2 = pq0, pq3 = pq0, pq4 = pq0;
    uint4 pq5 = make_uint4(0, 0, 0, 0);
    // ---- visits: the survivors in list order, constants through the scalar cache one visit ahead -------------------
    // One visit = our_gl.cpp:147-152 for one triangle on this block, one pixel per lane, up to the coverage decision.
    auto visit = [&](const TriScan& T, uint32_t tcur) {
        // the lanes of `m` note (u.x, u.y, triangle): moves under the mask, in place
        auto note = [&](unsigned long long m, double ux, double uy) {
            unsigned long long sv;
            asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, %[m]\n\tv_mov_b64 %[px], %[ux]\n\tv_mov_b64 %[py], %[uy]\n\t"
                         "v_mov_b32 %[pt], %[t]\n\tv_mov_b64 %[puz], %[uz]\n\tv_mov_b64 %[pruz], %[ruz]\n\t"
                         "v_mov_b64 %[pz0], %[z0]\n\tv_mov_b64 %[pz1], %[z1]\n\tv_mov_b64 %[pz2], %[z2]\n\tv_mov_b32 %[pcd], %[cd]\n\ts_mov_b64 exec, %[sv]"
                         : [px] "+v"(S.pux), [py] "+v"(S.puy), [pt] "+v"(S.ptri), [sv] "=&s"(sv),
                           [puz] "+v"(S.puz), [pruz] "+v"(S.pruz), [pz0] "+v"(S.pz0), [pz1] "+v"(S.pz1), [pz2] "+v"(S.pz2), [pcd] "+v"(S.pcd)
                         : [ux] "v"(ux), [uy] "v"(uy), [t] "s"(tcur), [m] "s"(m), [cd] "s"(KIND == TRGL_SHADER_FLAT ? T.color : T.dl),
                           [uz] "s"(T.uz), [ruz] "s"(T.ruz), [z0] "s"(T.z0), [z1] "s"(T.z1), [z2] "s"(T.z2));
        };
        double ux, uy;
        unsigned long long cm;                    // lanes with a covered pixel that can still win the z-test (wave-uniform)
        unsigned long long both, sv, sx;          // ... that still hold a fragment of an earlier triangle; scratch
#ifdef TRGL_DEBUG_COUNTERS
        unsigned long long am = 0;
#endif
        const uint32_t ext = T.by - T.bx;         // (hi >= lo in both 16-bit halves: no borrow)
        if (ALLWS || !(T.dl & TRGL_DL_LITERAL)) {
            // The core of a visit is written out instruction by instruction: 16 vector instructions, straight-line
            // but for two exits.  (From C++ the compiler builds it with exec-mask regions: zero-initialised phi registers,
            // save / restore pairs and a mask -> vector -> mask round trip for the ballot: ~30 vector + ~25 scalar.)  Every
            // vector instruction issues for all 64 lanes whether they matter or not, so nothing is masked: lanes outside
            // (bbox n block), hidden lanes and uncovered lanes only drop out of the final mask.
            //   NO test of the lane's pixel against the triangle's bbox (our_gl.cpp:130-133 only visits bbox pixels): a pixel centre
            //     outside floor(min) .. ceil(max) is at least half a pixel away from the triangle, and k_setup sends every triangle
            //     for which the roundings of u could bridge that (2^-40 S^2 R >= |u.z|) down the literal path below, which keeps the
            //     test.  Pixels beyond the image or outside the strip hold -inf and fail the depth comparison.
            //   barycentric(), our_gl.cpp:77-86: s0z = ax - x, s1z = ay - y (s0.xy, s1.xy and u.z hoisted into the record)
            //   depth first: zpl = fma(s0z, g1, fma(s1z, g2, c0)); a pixel with zpl >= its stored depth fails the z-test whatever
            //     its coverage (k_setup); NaN reads as keep.  No lane left: the visit ends here.
            //   u.x = s0y s1z - s0z s1y (geometry.h:145), u.y = s0z s1x - s0x s1z (:146), us = u.x + u.y, every product and sum
            //     rounded on its own as in the reference
            //   covered <=> !(us < u.z) && !(u.y > 0) && !(u.x > 0): u.z < 0 and nothing can over/underflow, so the signs of the
            //     quotients of :85 are known without dividing ("exactness"); max(u.x, u.y) > 0 <=> one of them is.
            double s0z, s1z, ta, tb2;
#ifdef TRGL_DEBUG_COUNTERS
            am = __ballot(__builtin_bit_cast(uint32_t, __builtin_elementwise_max(S.xy - __builtin_bit_cast(us2, T.bx), __builtin_bit_cast(us2, ext))) == ext);
#endif
            // (`asm goto` would let the exits of this block skip the test of `both` behind it; this compiler's AMDGPU back end
            // loses the block's contents when it has branch targets outside, so the exits set both = 0 and fall out.)
            asm volatile(
                "v_add_f64 %[s0z], %[ax], -%[pxc]\n\t"
                "v_add_f64 %[s1z], %[ay], -%[pyc]\n\t"
                "v_mov_b64 %[ta], %[c0]\n\t"
                "v_fmac_f64 %[ta], %[g2], %[s1z]\n\t"
                "v_fmac_f64 %[ta], %[g1], %[s0z]\n\t"
                "v_cmp_nge_f64_e32 vcc, %[ta], %[z]\n\t"
#ifdef TRGL_DEBUG_COUNTERS
                "s_mov_b64 %[cm], 0\n\t"
#endif
                "s_cbranch_vccz .Lvisit_end%=\n\t"
                "v_mul_f64 %[ta], %[s0y], %[s1z]\n\t"
                "v_mul_f64 %[tb], %[s1y], %[s0z]\n\t"
                "v_add_f64 %[ux], %[ta], -%[tb]\n\t"
                "v_mul_f64 %[ta], %[s1x], %[s0z]\n\t"
                "v_mul_f64 %[tb], %[s0x], %[s1z]\n\t"
                "v_add_f64 %[uy], %[ta], -%[tb]\n\t"
                "v_add_f64 %[ta], %[ux], %[uy]\n\t"
                "v_max_f64 %[tb], %[ux], %[uy]\n\t"
                "v_cmp_nlt_f64_e64 %[cm], %[ta], %[uz]\n\t"
                "s_and_b64 %[cm], %[cm], vcc\n\t"
                "v_cmp_nlt_f64_e32 vcc, 0, %[tb]\n\t"
                "s_and_b64 %[cm], %[cm], vcc\n\t"
                "s_cbranch_scc0 .Lvisit_end%=\n\t"
                // covered lanes: those that hold no fragment note (u.x, u.y, triangle) at once, in place (moves under the mask);
                // `both` = the covered lanes that still hold one (the code behind the block resolves first, then notes theirs)
                "s_andn2_b64 %[sv], %[cm], %[pend]\n\t"
                "s_cbranch_scc0 .Lvisit_noted%=\n\t"
                "s_mov_b64 %[sx], exec\n\t"
                "s_mov_b64 exec, %[sv]\n\t"
                "v_mov_b64 %[pux], %[ux]\n\t"
                "v_mov_b64 %[puy], %[uy]\n\t"
                "v_mov_b32 %[ptri], %[tcur]\n\t"
                "v_mov_b64 %[puz], %[uz]\n\t"
                "v_mov_b64 %[pruz], %[ruz]\n\t"
                "v_mov_b64 %[pz0], %[z0]\n\t"
                "v_mov_b64 %[pz1], %[z1]\n\t"
                "v_mov_b64 %[pz2], %[z2]\n\t"
                "v_mov_b32 %[pcd], %[cd]\n\t"
                "s_mov_b64 exec, %[sx]\n"
                ".Lvisit_noted%=:\n\t"
                "s_or_b64 %[pend], %[pend], %[cm]\n\t"
                "s_xor_b64 %[both], %[cm], %[sv]\n\t"
                "s_branch .Lvisit_done%=\n"
                ".Lvisit_end%=:\n\t"
                "s_mov_b64 %[both], 0\n"
                ".Lvisit_done%=:"
                : [cm] "=&s"(cm), [ux] "=&v"(ux), [uy] "=&v"(uy), [s0z] "=&v"(s0z), [s1z] "=&v"(s1z), [ta] "=&v"(ta), [tb] "=&v"(tb2),
                  [both] "=&s"(both), [sv] "=&s"(sv), [sx] "=&s"(sx), [pend] "+s"(pend), [pux] "+v"(S.pux), [puy] "+v"(S.puy), [ptri] "+v"(S.ptri),
                  [puz] "+v"(S.puz), [pruz] "+v"(S.pruz), [pz0] "+v"(S.pz0), [pz1] "+v"(S.pz1), [pz2] "+v"(S.pz2), [pcd] "+v"(S.pcd)
                : [pxc] "v"(S.pxc), [pyc] "v"(S.pyc), [z] "v"(S.z), [tcur] "s"(tcur),
                  [ax] "s"(T.ax), [ay] "s"(T.ay), [c0] "s"(T.c0), [g1] "s"(T.g1), [g2] "s"(T.g2),
                  [s0x] "s"(T.s0x), [s0y] "s"(T.s0y), [s1x] "s"(T.s1x), [s1y] "s"(T.s1y), [uz] "s"(T.uz),
                  [ruz] "s"(T.ruz), [z0] "s"(T.z0), [z1] "s"(T.z1), [z2] "s"(T.z2), [cd] "s"(KIND == TRGL_SHADER_FLAT ? T.color : T.dl)
                : "vcc", "scc");
        } else {
            // a triangle that is not well scaled: the same visit from C++, coverage from the literal quotients of :85
            const us2 off = S.xy - __builtin_bit_cast(us2, T.bx);
            const bool act = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(off, __builtin_bit_cast(us2, ext))) == ext;
            const double s0z = T.ax - S.pxc, s1z = T.ay - S.pyc;
            const double zpl = __builtin_fma(s0z, T.g1, __builtin_fma(s1z, T.g2, T.c0));
            ux = T.s0y * s1z - s0z * T.s1y;                               // geometry.h:145
            uy = s0z * T.s1x - T.s0x * s1z;                               // geometry.h:146
            const double us = ux + uy;
            const double b0 = 1.0 - us / T.uz, b1 = uy / T.uz, b2 = ux / T.uz;      // :85, as written
            cm = __ballot(act && !(zpl >= S.z) && !(b0 < 0 || b1 < 0 || b2 < 0));  // :152
#ifdef TRGL_DEBUG_COUNTERS
            am = __ballot(act);
#endif
            both = cm & pend;
            note(cm & ~pend, ux, uy);
            pend |= cm;
        }
        TRGL_DBG(2, tcur < fp.n_tris ? 1 : 0); TRGL_DBG(3, __popcll(am)); TRGL_DBG(4, cm ? 1 : 0); TRGL_DBG(5, __popcll(cm));
        if (both) {
            // Some covered lanes still hold a fragment of an earlier triangle, which has to be resolved first (submission
            // order per pixel).  Everything that can go goes in that one pass: the old fragments of all noted lanes AND the
            // new ones of the lanes that held none (`pend` holds them already); only the conflicting lanes' new fragments
            // stay noted afterwards.
            resolve<KIND, ALLWS, DEFERRED>(S, pend, recs, recs_w, draws, stats, zero_locked);
            note(both, ux, uy);
            asm volatile("s_mov_b64 %0, %1" : "+s"(pend) : "s"(both));      // pend = both, in the register that holds it
        }
    };
    const uint32_t p_first = beg & ~3u;
    auto load_step = [&](uint32_t p0, uint4& tri4, uint2& m) {
        const uint32_t pl = p0 + 4u * (uint32_t)lane, p = min(pl, (end - 1u) & ~3u);
        tri4 = *reinterpret_cast<const uint4*>(vals + p);
        m = *reinterpret_cast<const uint2*>(bmask + p);
    };
    uint4 tri_n = make_uint4(0, 0, 0, 0); uint2 m_n = make_uint2(0, 0);
    if (beg < end) load_step(p_first, tri_n, m_n);
    for (uint32_t p0 = p_first; p0 < end; p0 += 256) {
        const uint4 tri_c = tri_n; const uint2 m_c = m_n;
        TRGL_STAMP(10);            // startup (first step) / whatever is left between the stamps below
        load_step(p0 + 256, tri_n, m_n);                  // (past the end: clamped index; its entries are never looked at)
        // append the step's candidates to the ring in list order: entry 4 l + j comes after every entry of the lanes below l
        {
            // c_j: entry 4 l + j of the step reaches this wave's block (bit kblk of its mask) ...
            uint32_t c0 = (m_c.x >> kblk) & 1u, c1 = (m_c.x >> (16 + kblk)) & 1u, c2 = (m_c.y >> kblk) & 1u, c3 = (m_c.y >> (16 + kblk)) & 1u;
            if (p0 < beg || p0 + 256u > end) {
                // ... and lies in [beg, end): only the first and the last step of a list hold entries that do not.  Valid entries of
                // the lane: [lo, hi).  (Written without a wrapping subtraction under a select: `pl < end ? min(end - pl, 4) : 0` reached
                // the ISA as a saturating 4 - (end - pl), which takes every lane PAST the end for fully valid - lists shorter than a
                // step then appended whatever followed them in the pair buffer, stale words of an earlier flush included;
                // profiles/one_case.py with the diagnostic build found it.)
                const uint32_t pl = p0 + 4u * (uint32_t)lane;
                const uint32_t lo = min(beg - min(pl, beg), 4u), hi = min(end - min(pl, end), 4u);
                const uint32_t v = (0xfu << lo) & (0xfu >> (4u - hi));
                c0 &= v; c1 &= v >> 1; c2 &= v >> 2; c3 &= v >> 3;
            }
            if (fp.zq_cull) {
                // The flush holds large triangles: an entry whose depth bound (7 bits in its triangle word, k_setup) is not below the
                // block's largest stored depth cannot put a fragment into this block - it never becomes a candidate.  qb: the depth
                // rounded UP to the bound's grid (a float rounded up, then 2^-10 of a step more for the roundings here; 128 = nothing
                // stored yet).  Occluded triangles of frames with a high depth complexity die here, 256 entries per step, without
                // their records being touched.
                const float zmb = wave_max_f32(f32_up(S.z));
                const float tq = (zmb + 1.0f) * 64.0f + 0x1p-10f;
                const uint32_t qb = !(tq < 128.0f) ? 128u : (tq <= 1.0f ? 1u : (uint32_t)__builtin_ceilf(tq));      // (>= 1: zq = 0 says nothing)
                c0 = TRGL_VAL_ZQ(tri_c.x) < qb ? c0 : 0u; c1 = TRGL_VAL_ZQ(tri_c.y) < qb ? c1 : 0u;
                c2 = TRGL_VAL_ZQ(tri_c.z) < qb ? c2 : 0u; c3 = TRGL_VAL_ZQ(tri_c.w) < qb ? c3 : 0u;
            }
            const unsigned long long b0 = __ballot(c0), b1 = __ballot(c1), b2 = __ballot(c2), b3 = __ballot(c3);
            if (b0 | b1 | b2 | b3) {
                uint32_t pos = head + cnt;
                pos += __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u));
                pos += __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u));
                pos += __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, 0u));
                pos += __builtin_amdgcn_mbcnt_hi((uint32_t)(b3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b3, 0u));
                if (c0) ring[pos & (RING - 1)] = tri_c.x;
                pos += c0;
                if (c1) ring[pos & (RING - 1)] = tri_c.y;
                pos += c1;
                if (c2) ring[pos & (RING - 1)] = tri_c.z;
                pos += c2;
                if (c3) ring[pos & (RING - 1)] = tri_c.w;
#ifdef TRGL_DEBUG_COUNTERS
                {   // a candidate that is no triangle of the flush: remember where the first one came from
                    const uint32_t trs[4] = { tri_c.x, tri_c.y, tri_c.z, tri_c.w };
                    const uint32_t cs[4] = { c0, c1, c2, c3 };
                    for (int j = 0; j < 4; ++j)
                        if (cs[j] && TRGL_VAL_TRI(trs[j]) >= fp.n_tris && atomicCAS(&stats->dbg[10], 0ull, 1ull) == 0ull) {
                            stats->dbg[11] = ((unsigned long long)(p0 + 4u * (uint32_t)lane + (uint32_t)j) << 32) | trs[j];
                            stats->dbg[12] = ((unsigned long long)beg << 32) | end;
                            stats->dbg[13] = ((unsigned long long)(uint32_t)t << 32) | (uint32_t)kblk;
                        }
                }
#endif
                const uint32_t added = (uint32_t)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));
                cnt += added;
                TRGL_DBG(0, added);
            }
        }
        const bool last = p0 + 256 >= end;
        TRGL_STAMP(11);            // list step: wait for its entries, append the candidates
        // Rounds of up to 64 candidates, two in the pipe: the gather of round k + 1 is REQUESTED between the test of round k and the
        // visits of its survivors, so that it flies under them; its test follows the visits and sees the depths they left.  A
        // requested round also stays in flight over a list step when the ring cannot start the one after it yet.
        for (;;) {
            const bool can = cnt >= 64 || (last && cnt);
            bool keep = false;
            if (pn) {
                if (!can && !last) break;
            // ---- cull: lane = candidate ------------------------------------------------------------------------------
                // largest stored depth of the block (deferred fragments only lower depths later: a stale maximum stays a bound;
                // a pixel holding NaN can never be written again, v_max skips it)
                // (as a float rounded up: six v_max_f32 with DPP / permlane operands instead of 64-bit shuffles through LDS)
                const double zmaxb = (double)wave_max_f32(f32_up(S.z));
                keep = (uint32_t)lane < pn;
#ifdef TRGL_DEBUG_COUNTERS
                if (keep && ptri >= fp.n_tris) { atomicAdd(&stats->dbg[9], 1ull); keep = false; }      // must stay 0: a list entry that is no triangle
#endif
                if (keep) {
                    const double2 q0 = pq0, q1 = pq1, q2 = pq2, q3 = pq3, q4 = pq4;
                    const uint4 q5 = pq5;
                    if (ALLWS || !(q5.w & TRGL_DL_LITERAL)) {
                        const double ax = q0.x, ay = q0.y, s0x = q1.x, s0y = q1.y, s1x = q2.x, s1y = q2.y, c0 = q3.x, uz = q3.y, g1 = q4.x, g2 = q4.y;
                        // pixel centres of (bbox n block): the rectangle with centre (xm, ym) and half-widths (xh, yh)
                        const int bx0 = (int)(q5.x & 0xffffu), by0 = (int)(q5.x >> 16), bx1 = (int)(q5.y & 0xffffu), by1 = (int)(q5.y >> 16);
                        const int xl = max(bx0, X0), xr = min(bx1, X0 + 7), yl = max(by0, Y0), yr = min(by1, Y0 + 7);
                        const double xm = __builtin_fma((double)(xl + xr), 0.5, 0.5), xh = (double)(xr - xl) * 0.5;
                        const double ym = __builtin_fma((double)(yl + yr), 0.5, 0.5), yh = (double)(yr - yl) * 0.5;
                        const double dxm = ax - xm, dym = ay - ym;
                        // (edge)  a pixel is covered iff the rounded u.x <= 0, u.y <= 0 and u.x + u.y >= u.z (the sign form of :152).
                        // Each is affine in the pixel centre: over the rectangle it stays within (sum of |gradient| x half-width) of its
                        // value at the centre, and the value computed THERE (with the operations of the scan) differs from what the
                        // scan computes at a pixel by that affine change + at most 2^-51 (sum of |products|) <= 2^-51 Sa R:
                        // margins of 2^-40 of the same magnitudes.  (No corner is selected: a select of a double is two instructions.)
                        const double uxc = s0y * dym - dxm * s1y;                                             // u.x at the centre
                        const double uyc = dxm * s1x - s0x * dym;                                             // u.y
                        const double gx = s1y - s1x, gy = s0x - s0y;                                           // gradient of u.x + u.y
                        const double sp1 = __builtin_fma(fabs(s1y), xh, fabs(s0y) * yh);                      // how far u.x can fall below uxc
                        const double sp2 = __builtin_fma(fabs(s1x), xh, fabs(s0x) * yh);                      // ... u.y below uyc
                        const double sp3 = __builtin_fma(fabs(gx), xh, fabs(gy) * yh);                        // ... u.x + u.y rise above uxc + uyc
                        const double R = ((fabs(dxm) + xh) + (fabs(dym) + yh)) + 16.0;                       // >= |A - pixel| (L1) on the rectangle
                        const double ma = 0x1p-40 * ((fabs(s0y) + fabs(s1y)) * R) + 0x1p-1000;
                        const double mb = 0x1p-40 * ((fabs(s0x) + fabs(s1x)) * R) + 0x1p-1000;
                        // (depth)  every covered pixel's z is above the plane c0 + (ax - x) g1 + (ay - y) g2 (k_setup), whose minimum
                        // over the rectangle is its value at the centre minus (|g1| xh + |g2| yh); if even that is not below the block's
                        // largest stored depth, no pixel of the block passes the strict `<` of :165.  NaN compares false = keep.
                        // The plane reaches below the triangle's own depths outside the triangle; a covered pixel also has b_i >= 0 and
                        // b0 + b1 + b2 = 1 +- 2^-50, hence z >= min(z0, z1, z2) - 2^-40 max|z_i|, and the plane's values AT the three vertices
                        // A, A + (s0y, s1y), A + (s0x, s1x) are z_i minus k_setup's margin (>= 2^-39 max|z_i| (R S/|u.z| + 1) with R >= S, which
                        // also covers the roundings of these evaluations): their minimum is below that bound.  The larger of the two
                        // lower bounds counts.
                        const double zpl = __builtin_fma(dxm, g1, __builtin_fma(dym, g2, c0)) - __builtin_fma(fabs(g1), xh, fabs(g2) * yh);
                        const double zv = vmin(vmin(c0, __builtin_fma(-s0y, g1, __builtin_fma(-s1y, g2, c0))), __builtin_fma(-s0x, g1, __builtin_fma(-s1x, g2, c0)));
                        if (uxc - sp1 > ma || uyc - sp2 > mb || (uxc + uyc) + sp3 < uz - (ma + mb) || vmax(zpl, zv) >= zmaxb) keep = false;
                    }
                }
            } else if (!can) break;
            const unsigned long long surv = __ballot(keep);
            const uint32_t tri = ptri;                   // the tested round's candidates (the survivors among them are visited below)
            TRGL_DBG(1, __popcll(surv));
            TRGL_STAMP(12);        // cull round (tests)
            if (can) {
                const uint32_t n = cnt < 64 ? cnt : 64;
                __builtin_amdgcn_wave_barrier();
                ptri = TRGL_VAL_TRI(ring[(head + lane) & (RING - 1)]);
                __builtin_amdgcn_wave_barrier();
                head += n; cnt -= n; pn = n;
                if ((uint32_t)lane < n
#ifdef TRGL_DEBUG_COUNTERS
                    && ptri < fp.n_tris
#endif
                    ) {
                    const double2* q = reinterpret_cast<const double2*>(recs + ptri);
                    pq0 = q[0]; pq1 = q[1]; pq2 = q[2]; pq3 = q[3]; pq4 = q[4];
                    pq5 = reinterpret_cast<const uint4*>(q)[7];
                }
            } else pn = 0;
            if (surv) {
                // The survivors move to lanes 0 .. ns - 1 (through LDS, order kept), so that the visit loop counts instead of peeling bits
                // off a 64-bit mask: 2 scalar instructions per visit instead of 8 - the scalar unit, one instruction per cycle for the
                // whole CU, is as busy in this kernel as the vector units.
                const uint32_t ns = (uint32_t)__popcll(surv);
                uint32_t* comp = s_out[w];
                if (keep) comp[__builtin_amdgcn_mbcnt_hi((uint32_t)(surv >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)surv, 0u))] = tri;
                __builtin_amdgcn_wave_barrier();
                uint32_t tri_s = comp[lane];
                // The lanes behind them name the record behind the flush's last one (k_make_items): the loop below runs in PAIRS of visits -
                // one counter update and one exit test per pair - and the partner of an odd last survivor is that record, whose
                // visit ends at its first test.
                if ((uint32_t)lane >= ns) tri_s = fp.n_tris;
                // Two sets of constants, A and B, take turns (no register copies): while a visit works on one, the request for
                // the next survivor's record fills the other.  The wait for a set sits behind the visit that ran under its request.
                // (The last request of a round names lane ns + 1 or, behind lane 63, lane 0 or 1: valid addresses, nobody reads the result.)
                uint32_t i = 0;
                uint32_t ta = (uint32_t)__builtin_amdgcn_readlane((int)tri_s, 0), tb;
                TriScan A = load_scan(recs, ta), B;
                scan_wait(A);                        // the first triangle's constants are waited for HERE, not behind the first request inside the loop
                do {
                    tb = (uint32_t)__builtin_amdgcn_readlane((int)tri_s, (int)(i + 1u));
                    B = load_scan(recs, tb);
                    __builtin_amdgcn_sched_barrier(0);
                    visit(A, ta);
                    __builtin_amdgcn_sched_barrier(0);
                    scan_wait(B);
                    ta = (uint32_t)__builtin_amdgcn_readlane((int)tri_s, (int)(i + 2u));
                    A = load_scan(recs, ta);
                    __builtin_amdgcn_sched_barrier(0);
                    visit(B, tb);
                    __builtin_amdgcn_sched_barrier(0);
                    scan_wait(A);
                    i += 2;
                } while (i < ns);
            }
            TRGL_STAMP(13);        // visits of the round (resolves included; their own clock is counter 14)
        }
    }
    if (pend) resolve<KIND, ALLWS, DEFERRED>(S, pend, recs, recs_w, draws, stats, zero_locked);

    TRGL_STAMP(13);
    // ---- block out: every owned pixel once ---------------------------------------------------------------------------
    // depth: 8 B per lane, a row of the block is 64 contiguous bytes
    if (owned) __builtin_nontemporal_store(S.z, &fp.zb[pix]);
    if (DEFERRED && fp.idbuf && owned) fp.idbuf[pix] = S.id;
    {
        const int bpp = BPP ? BPP : fp.bpp;
        const bool whole = fp.init_from_clear && X0 + 7 < fp.W;      // every pixel of an owned row is written: rows as dwords
        if (bpp == 4) {
            if (owned && (fp.init_from_clear || S.frags)) *reinterpret_cast<uint32_t*>(fp.fb + pix * 4) = S.color;
        } else if (bpp == 3 && whole) {
            // a row of the block is 24 B = 6 dwords: lanes 8 r + 0..5 assemble them from the row's 8 colours (through LDS)
            uint32_t* so = s_out[w];
            so[lane] = S.color;
            __builtin_amdgcn_wave_barrier();
            const int r = lane >> 3, dw = lane & 7;
            if (dw < 6) {
                const int y = Y0 + r;
                if (y < fp.H && y >= fp.strip_y0 && y < fp.strip_y1) {
                    // bytes 4 dw .. 4 dw + 3 of the row: pixel (4 dw + i) / 3, channel (4 dw + i) % 3
                    const int p0 = (4 * dw) / 3, sh = (4 * dw) % 3;
                    // (two pixels hold them: 3 - sh bytes of pixel p0, the rest of pixel p0 + 1 <= 7)
                    const uint32_t ca = so[8 * r + p0] & 0xffffffu, cb2 = so[8 * r + p0 + 1] & 0xffffffu;
                    const unsigned long long bytes = ((unsigned long long)ca | ((unsigned long long)cb2 << 24)) >> (8 * sh);
                    uint32_t* dst = reinterpret_cast<uint32_t*>(fp.fb + ((size_t)X0 + (size_t)y * fp.W) * 3) + dw;
                    *dst = (uint32_t)bytes;
                }
            }
        } else {
            if (owned && (fp.init_from_clear || S.frags)) {
                uint8_t* dst = fp.fb + pix * bpp;
                if (bpp == 3) { *reinterpret_cast<uint16_t*>(dst) = (uint16_t)S.color; dst[2] = (uint8_t)(S.color >> 16); }   // TGAImage::set: b, g, r
                else dst[0] = (uint8_t)S.color;
            }
        }
    }

    TRGL_STAMP(15);                // block out
    // ---- stats: our_gl.cpp:194-198, reduced per wave, then per workgroup: one partial per work item (k_fold_stats) ------
    // (std::min over the written depths of a pixel is its last one: the z-test only lets smaller ones through)
    // One reduction per WORKGROUP: every lane leaves its three values in its wave's candidate ring (done with by now), and behind the
    // kernel's only block-level barrier the first wave folds the four waves' values lane by lane and runs the butterflies once
    // (~100 vector + 30 LDS instructions that three of four waves no longer spend; three same-address LDS atomics per wave instead
    // cost 0.5 ms of k_raster - the LDS serialises the 64 lanes).
    {
        unsigned long long* red = reinterpret_cast<unsigned long long*>(ring);       // [0..63] smallest, [64..127] largest depth key, then the counts
        red[lane] = zkey(S.frags ? S.z : __builtin_inf());
        red[64 + lane] = zkey(S.zmax);
        ring[256 + lane] = S.frags;
    }
#ifdef TRGL_DEBUG_COUNTERS
    for (int o = 32; o; o >>= 1) S.dbg[8] += __shfl_xor(S.dbg[8], o);
    if (lane == 0) for (int k = 0; k < 16; ++k) if (S.dbg[k]) atomicAdd(&stats->dbg[k], S.dbg[k]);
#endif
    __syncthreads();                 // the only block-level barrier of the kernel, when every wave of the workgroup is done
    if (w == 0) {
        unsigned long long frags = 0, kmin = ~0ull, kmax = 0ull;
        for (int k = 0; k < 4; ++k) {
            const unsigned long long* red = reinterpret_cast<const unsigned long long*>(s_ring[k]);
            const unsigned long long a = red[lane], b = red[64 + lane];
            frags += s_ring[k][256 + lane]; kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax;
        }
        for (int o = 32; o; o >>= 1) {
            frags += __shfl_xor(frags, o);
            unsigned long long a = __shfl_xor(kmin, o); kmin = a < kmin ? a : kmin;
            unsigned long long b = __shfl_xor(kmax, o); kmax = b > kmax ? b : kmax;
        }
        if (lane == 0) {
            ulonglong2* dst = reinterpret_cast<ulonglong2*>(my_stats);
            dst[0] = make_ulonglong2(frags, kmin);
            dst[1] = make_ulonglong2(kmax, 0ull);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_shade<PHONG|EYE>: the fragment stage of a PHONG / EYE flush, once per visible pixel.
// k_raster left, for every pixel whose depth it wrote in this flush, the record index of the LAST triangle that passed the
// z-test there (fp.idbuf).  IShader::fragment has no side effects and always returns discard = false (main.cpp:92-170,
// 220-261), so calling it for that triangle only gives the same framebuffer as calling it for every z-pass in order
// (our_gl.cpp:187-192) - with 64 busy lanes per wave instead of the few pixels of one small triangle.  One 256-thread
// block per work item of k_raster (so tiles this flush did not touch are not visited), one 8x8 block per wave; the
// barycentrics are recomputed per pixel with exactly the operations of the scan (same bits).
// ---------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_shade(FrameParams fp, const TriRec* __restrict__ recs, const TriW* __restrict__ recs_w,
                                                const DrawDesc* __restrict__ draws,
                                                const DevTexture* __restrict__ tex, const uint4* __restrict__ items,
                                                const uint32_t* __restrict__ n_items) {
    const int lane = threadIdx.x & 63;
    const uint32_t item_idx = blockIdx.x;                // one 256-thread block per work item: wave w shades block w of the item's row
    if (item_idx >= *n_items) return;
    const uint32_t item = items[item_idx].x;
    if (item & TRGL_ITEM_CLEAR) return;                  // no triangles: no owners
    const int t = (int)(item & 0xffffffu);
    const int tile_y = t / fp.tiles_x, tile_x = t - tile_y * fp.tiles_x;
    const int x = (tile_x << TRGL_TILE_LOG2) + 8 * (int)(threadIdx.x >> 6) + (lane & 7);
    const int y = (tile_y << TRGL_TILE_LOG2) + 8 * (int)((item >> 24) & 3u) + (lane >> 3);
    const bool mine = x < fp.W && y < fp.H && y >= fp.strip_y0 && y < fp.strip_y1;
    const size_t idx = (size_t)x + (size_t)y * fp.W;
    // The kernel is bound by dependent memory latencies: owner -> record -> varyings -> texels.  The visibility buffer holds
    // `draw << 24 | triangle in its draw`, from which BOTH the record (recs[draw.first + triangle]) and the varyings are addressed:
    // two loads in flight instead of one after the other.
    const uint32_t dl = mine ? fp.idbuf[idx] : 0xffffffffu;
    if (dl == 0xffffffffu) return;
    // the draw (uniforms, varyings array, first record) is wave-uniform in all but exotic flushes: serve one draw at a time
    uint32_t color = 0;
    unsigned long long todo = __ballot(true);
    while (todo) {
        const int src = __builtin_ctzll(todo);
        const uint32_t di = (uint32_t)__builtin_amdgcn_readlane((int)(dl >> 24), src);
        const bool here = (dl >> 24) == di;
        todo &= ~__ballot(here);
        if (here) {
            // descriptors through the CONSTANT address space: the draw index is wave-uniform, so every field (uniforms, texture
            // descriptors) comes in by scalar loads through the scalar cache.  As plain global loads they were vector loads, each
            // waited for on its own: ~20 memory round trips in a row per block, 68 % of the kernel's wave time in s_waitcnt.
            typedef const __attribute__((address_space(4))) DrawDesc CDraw;
            typedef const __attribute__((address_space(4))) DevTexture CTex;
            // (under `here` the compiler knows di == dl >> 24 and would address the draw per lane: re-derive it as a scalar)
            const uint32_t di_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(dl >> 24));
            CDraw& d = ((CDraw*)draws)[di_s];
            CTex* const ctex = (CTex*)tex;
            const uint32_t local = dl & 0xffffffu;
            const TriRec& r = recs[d.first + local];
            const TriW& rw = recs_w[d.first + local];
            const double* vary = d.vary + (size_t)local * 24;
            // barycentric(), our_gl.cpp:77-86, as in k_raster
            const double pxc = (double)x + 0.5, pyc = (double)y + 0.5;
            const double s0z = r.ax - pxc, s1z = r.ay - pyc;
            const double ux = r.s0y * s1z - s0z * r.s1y;
            const double uy = s0z * r.s1x - r.s0x * s1z;
            const double us = ux + uy;
            double b0, b1, b2;
            if (r.ruz != 0.0) {
                b0 = 1.0 - div_by_uz(us, r.uz, r.ruz); b1 = div_by_uz(uy, r.uz, r.ruz); b2 = div_by_uz(ux, r.uz, r.ruz);
            } else {
                b0 = 1.0 - us / r.uz; b1 = uy / r.uz; b2 = ux / r.uz;
            }
            double pc[3];
            const double denom = b0 * rw.iw0 + b1 * rw.iw1 + b2 * rw.iw2;                     // our_gl.cpp:172-174
            if (fabs(denom) < 1e-15) { pc[0] = b0; pc[1] = b1; pc[2] = b2; }                  // :177-185
            else { pc[0] = (b0 * rw.iw0) / denom; pc[1] = (b1 * rw.iw1) / denom; pc[2] = (b2 * rw.iw2) / denom; }
            const int kind = KIND == KIND_ANY ? d.kind : KIND;        // wave-uniform inside this iteration
            color = kind == TRGL_SHADER_PHONG ? frag_phong(d.u, ctex, vary, pc).bgra : frag_eye(d.u, ctex, vary, pc).bgra;
        }
    }
    uint8_t* dst = fp.fb + idx * fp.bpp;                                               // TGAImage::set, tgaimage.cpp:32-39
    if (fp.bpp == 3) { dst[0] = (uint8_t)color; dst[1] = (uint8_t)(color >> 8); dst[2] = (uint8_t)(color >> 16); }
    else if (fp.bpp == 4) *reinterpret_cast<uint32_t*>(dst) = color;
    else dst[0] = (uint8_t)color;                                  // bpp is 1, 3 or 4 (trgl_create)
}

// ---- self-test of the two exactness shortcuts, against the hardware's IEEE division ------------------
__device__ __forceinline__ unsigned long long sm64(unsigned long long& st) {
    unsigned long long z = (st += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double mk_double(unsigned long long mant, int exp2, bool neg) {
    unsigned long long bits = (mant & 0x000fffffffffffffull) | ((unsigned long long)(exp2 + 1023) << 52) | (neg ? 0x8000000000000000ull : 0ull);
    return __longlong_as_double((long long)bits);
}
__global__ void k_selftest_division(unsigned long long n_per_thread, unsigned long long seed, unsigned long long* mismatches) {
    unsigned long long st = seed + 0x1000003ull * (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x);
    unsigned long long bad = 0;
    for (unsigned long long it = 0; it < n_per_thread; ++it) {
        unsigned long long r0 = sm64(st), r1 = sm64(st), r2 = sm64(st);
        int mode = (int)(r2 & 7);
        int eb = (int)((r2 >> 8) % 440) - 40;           // |uz| in [2^-40, 2^400): the well-scaled range
        int ea = (int)((r2 >> 24) % 700) - 350;
        unsigned long long mb = r1;
        if (mode == 1) mb = 0x000fffffffffffffull;                       // all-ones significand
        if (mode == 2) mb = 0x000fffffffffffffull ^ (1ull << (r1 % 52));  // one zero bit
        if (mode == 3) mb = 0;                                           // power of two
        double uz = mk_double(mb, eb, true);                             // u.z < 0
        double a = mk_double(r0, ea, (r2 >> 40) & 1);
        if (mode >= 4 && mode <= 6) {
            // numerator chosen so that a/uz lies next to a rounding midpoint: a = RN(uz * (q + ulp(q)/2 * (1 +- tiny)))
            double q = mk_double(r0, ea - eb, (r2 >> 41) & 1);
            double half_ulp = mk_double(0, ea - eb - 53, false);
            a = __builtin_fma(uz, q, uz * half_ulp);
            if (mode == 5) a = __longlong_as_double(__double_as_longlong(a) + 1);
            if (mode == 6) a = __longlong_as_double(__double_as_longlong(a) - 1);
        }
        if (mode == 7) a = (r2 >> 42) & 1 ? 0.0 : -0.0;
        double ruz = 1.0 / uz;
        double ref = a / uz, got = div_by_uz(a, uz, ruz);
        if (__double_as_longlong(ref) != __double_as_longlong(got)) ++bad;
        // sign shortcuts of the coverage test
        if ((ref < 0) != (a > 0.0)) ++bad;
        if (((1.0 - ref) < 0) != (a < uz)) ++bad;
        // numerator right next to uz, where 1 - q changes sign
        double a2 = __longlong_as_double(__double_as_longlong(uz) + (long long)(r1 % 5) - 2);
        double ref2 = a2 / uz;
        if (__double_as_longlong(ref2) != __double_as_longlong(div_by_uz(a2, uz, ruz))) ++bad;
        if (((1.0 - ref2) < 0) != (a2 < uz)) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// ---- self-test of the samplers' index math: tex_fetch on given uv (the fixtures come from the compiled reference's
// IShader::sample2D, tests/golden/sampler_golden.npz) ---------------------------------------------------------------
__global__ void k_selftest_sampler(const DevTexture* __restrict__ tex, int slot, const double* __restrict__ uv, unsigned long long n,
                                   uint8_t* __restrict__ out) {
    const unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DevTexture* t = tex_slot(tex, slot);
    Color c = t ? tex_fetch(t, uv + 2 * i) : Color{ 0xffffffffu, 4 };        // a missing map samples as opaque white (model.cpp:416-418)
    for (int k = 0; k < 4; ++k) out[5 * i + k] = (uint8_t)(c.bgra >> (8 * k));
    out[5 * i + 4] = (uint8_t)c.bytespp;
}

// Work items of the raster kernel: one per workgroup = one row of four 8x8 blocks of a tile with triangles (four items
// per tile), or one whole tile that is only cleared.  Tiles outside the context's tile rows, empty tiles of a flush that
// does not start from clear, and block rows entirely outside the strip get no item.
__global__ __launch_bounds__(256) void k_make_items(FrameParams fp, const uint32_t* __restrict__ tile_start,
                                                    const uint32_t* __restrict__ tile_end,
                                                    uint4* __restrict__ items, uint32_t* __restrict__ n_items, TriRec* __restrict__ recs) {
    const int ntiles_strip = (fp.strip_ty1 - fp.strip_ty0) * fp.tiles_x;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    // The record behind the flush's last one belongs to no triangle: k_raster visits it when a cull round leaves an odd number of
    // survivors (its visit loop runs in pairs).  Its depth plane is +inf everywhere, so no pixel is in front of its stored depth and
    // the visit ends at its first test; should a pixel hold NaN (which lets every plane through), u.z = 1 > 0 = u.x + u.y fails coverage.
    if (k == 0 && fp.n_tris) {
        TriRec d;
        d.ax = d.ay = d.s0x = d.s0y = d.s1x = d.s1y = 0.0; d.c0 = __builtin_inf(); d.uz = 1.0; d.g1 = d.g2 = 0.0;
        d.ruz = 0.0; d.z0 = d.z1 = d.z2 = 0.0; d.bx0 = d.by0 = 1; d.bx1 = d.by1 = 0; d.color = 0; d.dl = 0;
        recs[fp.n_tris] = d;
    }
    uint32_t nb = 0, t = 0, rows = 0;          // rows: bit r set = block row r of the tile gets an item
    uint32_t beg = 0, end = 0;
    bool clear_only = false;
    if (k < ntiles_strip) {
        t = (uint32_t)(fp.strip_ty0 * fp.tiles_x + k);
        beg = tile_start[t]; end = tile_end[t];
        const uint32_t n = end - beg;
        const int ty = (int)(t / (uint32_t)fp.tiles_x);
        if ((n || fp.init_from_clear) && tile_row_owned(fp, ty)) {
            if (n) {
                for (int r = 0; r < 4; ++r) {
                    const int y0 = ty * TRGL_TILE + 8 * r, y1 = y0 + 7;
                    if (y0 < fp.H && y1 >= fp.strip_y0 && y0 < fp.strip_y1) rows |= 1u << r;
                }
                nb = (uint32_t)__popc(rows);
            } else { clear_only = true; nb = 1; }
        }
    }
    // wave-aggregated append (order of items is irrelevant)
    uint32_t inc = nb;
    const int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if (lane >= o) inc += v; }
    const uint32_t total = __shfl(inc, 63);
    uint32_t base = 0;
    if (lane == 63 && total) base = atomicAdd(n_items, total);
    base = __shfl(base, 63) + inc - nb;
    // (an item carries its tile's slice of the pair list: one dependent load less at the start of every raster wave)
    if (clear_only) items[base] = make_uint4(t | TRGL_ITEM_CLEAR, 0u, 0u, 0u);
    else for (uint32_t r = 0; r < 4; ++r) if ((rows >> r) & 1u) items[base++] = make_uint4(t | (r << 24), beg, end, 0u);
}

// after the raster kernel of a flush: fold the per-item partials into the context's counters (our_gl.cpp:194-198) and fix the
// sign of a zero z-range end (see DevStats).  FOLD_BLOCKS blocks take a slice each and add it with three atomics; the block
// that finishes last (a counter behind a device-scope fence) settles the per-flush state.  (One block walking 65536 partials
// took 39 us of a 2.9 ms frame.)
constexpr int FOLD_BLOCKS = 32;
__global__ __launch_bounds__(1024) void k_fold_stats(DevStats* __restrict__ s, uint32_t* __restrict__ n_items,
                                                     const unsigned long long* __restrict__ item_stats) {
    __shared__ unsigned long long sh[3][16];
    __shared__ bool s_last;
    const uint32_t n = *n_items;
    unsigned long long fr = 0, kmin = ~0ull, kmax = 0ull;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const ulonglong2 a = reinterpret_cast<const ulonglong2*>(item_stats + (size_t)i * 4)[0];
        const unsigned long long c = item_stats[(size_t)i * 4 + 2];
        fr += a.x; kmin = a.y < kmin ? a.y : kmin; kmax = c > kmax ? c : kmax;
    }
    for (int o = 32; o; o >>= 1) {
        fr += __shfl_xor(fr, o);
        unsigned long long a = __shfl_xor(kmin, o); kmin = a < kmin ? a : kmin;
        unsigned long long b = __shfl_xor(kmax, o); kmax = b > kmax ? b : kmax;
    }
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sh[0][w] = fr; sh[1][w] = kmin; sh[2][w] = kmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k) {
            fr += sh[0][k]; kmin = sh[1][k] < kmin ? sh[1][k] : kmin; kmax = sh[2][k] > kmax ? sh[2][k] : kmax;
        }
        if (fr) {       // items without fragments carry the neutral keys
            atomicAdd(&s->fragments, fr);
            atomicMin(&s->zmin_key, kmin);
            atomicMax(&s->zmax_key, kmax);
        }
        __threadfence();
        s_last = atomicAdd(&n_items[1], 1u) == gridDim.x - 1;       // n_items[1]: blocks of this launch that are done
    }
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        __threadfence();
        if (!s->zero_locked && (s->zero_pos_key != TRGL_ZERO_KEY_EMPTY || s->zero_neg_key != TRGL_ZERO_KEY_EMPTY)) {
            s->zero_sign = s->zero_neg_key < s->zero_pos_key ? 1u : 0u;
            s->zero_locked = 1u;
        }
        s->literal_tris = 0; s->large_tris = 0;    // counted per flush (k_setup)
        s->zero_pos_key = TRGL_ZERO_KEY_EMPTY;
        s->zero_neg_key = TRGL_ZERO_KEY_EMPTY;
        n_items[0] = 0;         // every block read it before its atomic above; k_make_items of the next flush appends from 0
        n_items[1] = 0;
    }
}

}  // namespace

namespace trgl {

void launch_selftest_division(hipStream_t s, unsigned long long n_per_thread, unsigned long long seed, unsigned long long* mismatches) {
    hipLaunchKernelGGL(k_selftest_division, dim3(1024), dim3(256), 0, s, n_per_thread, seed, mismatches);
}

void launch_selftest_sampler(hipStream_t s, const DevTexture* tex, int slot, const double* uv, unsigned long long n, uint8_t* out) {
    if (n) hipLaunchKernelGGL(k_selftest_sampler, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tex, slot, uv, n, out);
}

uint32_t owned_tiles(const FrameParams& fp) {
    if (fp.il_tiles == 0) return (uint32_t)((fp.strip_ty1 - fp.strip_ty0) * fp.tiles_x);
    return (uint32_t)(il_owned_below(fp, fp.tiles_y) * fp.tiles_x);
}

// at most four work items (rows of blocks) per owned tile
uint32_t raster_max_items(const FrameParams& fp) { return owned_tiles(fp) * 4u; }

void launch_raster(hipStream_t s, const FrameParams& fp, int kind /* TRGL_SHADER_* if uniform over the flush, else -1 */, bool all_well_scaled,
                   const TriRec* recs, const TriW* recs_w, const uint32_t* vals, const uint16_t* bmask,
                   const uint32_t* tile_start, const uint32_t* tile_end, const DrawDesc* draws,
                   const DevTexture* tex, DevStats* stats, uint32_t max_items, uint4* items,
                   uint32_t* n_items, unsigned long long* item_stats, hipEvent_t ev_before, hipEvent_t ev_after) {
    const int tiles = (fp.strip_ty1 - fp.strip_ty0) * fp.tiles_x;
    if (tiles <= 0 || max_items == 0) {          // a context that owns no rows (a rank beyond the image's bands): nothing to draw
        if (ev_before) (void)hipEventRecord(ev_before, s);
        if (ev_after) (void)hipEventRecord(ev_after, s);
        hipLaunchKernelGGL(k_fold_stats, dim3(FOLD_BLOCKS), dim3(1024), 0, s, stats, n_items, item_stats);     // n_items is 0: only resets the per-flush counts
        return;
    }
    hipLaunchKernelGGL(k_make_items, dim3((tiles + 255) / 256), dim3(256), 0, s, fp, tile_start, tile_end, items, n_items, const_cast<TriRec*>(recs));
    dim3 grid(((max_items + 7u) / 8u) * 8u);    // workgroup b -> item (b mod 8) * ceil(G / 8) + b / 8 (k_raster)
    if (ev_before) (void)hipEventRecord(ev_before, s);
#define TRGL_LAUNCH_RASTER(...) hipLaunchKernelGGL((k_raster<__VA_ARGS__>), grid, dim3(256), 0, s, fp, recs, recs_w, vals, bmask, tile_start, tile_end, draws, stats, items, n_items, item_stats)
    switch (kind) {
    case TRGL_SHADER_FLAT:
        if (all_well_scaled) {
            if (fp.bpp == 3) TRGL_LAUNCH_RASTER(TRGL_SHADER_FLAT, 3, true);
            else if (fp.bpp == 4) TRGL_LAUNCH_RASTER(TRGL_SHADER_FLAT, 4, true);
            else TRGL_LAUNCH_RASTER(TRGL_SHADER_FLAT, 0, true);
        } else {
            TRGL_LAUNCH_RASTER(TRGL_SHADER_FLAT);
        }
        break;
    case TRGL_SHADER_GOURAUD: if (all_well_scaled) TRGL_LAUNCH_RASTER(TRGL_SHADER_GOURAUD, 0, true); else TRGL_LAUNCH_RASTER(TRGL_SHADER_GOURAUD); break;
    case TRGL_SHADER_PHONG:   if (all_well_scaled) TRGL_LAUNCH_RASTER(TRGL_SHADER_PHONG, 0, true); else TRGL_LAUNCH_RASTER(TRGL_SHADER_PHONG); break;
    case TRGL_SHADER_EYE:     if (all_well_scaled) TRGL_LAUNCH_RASTER(TRGL_SHADER_EYE, 0, true); else TRGL_LAUNCH_RASTER(TRGL_SHADER_EYE); break;
    case TRGL_SHADER_CHECKER: TRGL_LAUNCH_RASTER(TRGL_SHADER_CHECKER); break;
    default:                  TRGL_LAUNCH_RASTER(KIND_ANY); break;
    }
#undef TRGL_LAUNCH_RASTER
    if (ev_after) (void)hipEventRecord(ev_after, s);
    if (fp.idbuf) {                                     // the flush has PHONG / EYE draws: shade the visible pixels they own
#define TRGL_LAUNCH_SHADE(K) hipLaunchKernelGGL(k_shade<K>, dim3(max_items), dim3(256), 0, s, fp, recs, recs_w, draws, tex, items, n_items)
        if (kind == TRGL_SHADER_PHONG) TRGL_LAUNCH_SHADE(TRGL_SHADER_PHONG);
        else if (kind == TRGL_SHADER_EYE) TRGL_LAUNCH_SHADE(TRGL_SHADER_EYE);
        else TRGL_LAUNCH_SHADE(KIND_ANY);
#undef TRGL_LAUNCH_SHADE
    }
    hipLaunchKernelGGL(k_fold_stats, dim3(FOLD_BLOCKS), dim3(1024), 0, s, stats, n_items, item_stats);
}

}  // namespace trgl
