// kernels_raster.hip — the tile rasterizer (gfx950): bounding-box scan, barycentric coverage test,
// fp64 z-test and the fragment shader of our_gl.cpp:147-199, one WAVEFRONT per 32x32 screen tile.
//
// The tile's z-buffer (fp64) lives in LDS for the whole tile, so the compare-and-write of
// our_gl.cpp:165,191 needs no atomics: one wave walks the tile's triangle list in submission order
// and, within one triangle, every lane owns a different pixel.  Depths leave the chip once, with
// row-contiguous stores; colours go straight to the framebuffer as fragments pass (FLAT, GOURAUD)
// or are shaded once per visible pixel afterwards (PHONG, EYE: k_shade).
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
// v_min3_f32 / v_max_f32 as single instructions (a NaN operand is skipped)
__device__ __forceinline__ float fmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float fmax2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
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

// LDS index of pixel (x,y) of the tile: rows are XOR-swizzled in 8-pixel groups so that the 8x8
// pixel block a wave touches per step (4 rows per 32-lane group) is bank-conflict free for both the
// 8-byte z reads and the 4-byte colour reads.
__device__ __forceinline__ int lds_index(int x, int y) {
    return ((y & (TRGL_TILE - 1)) << TRGL_TILE_LOG2) + ((x & (TRGL_TILE - 1)) ^ ((y & 3) << 3));
}

// A wave's batch of triangle records: lane i holds record i (8 x 16 B) and its triangle id.
struct RecQ { uint4 q[8]; uint32_t tri; };
__device__ __forceinline__ RecQ load_rec(const TriRec* __restrict__ recs, uint32_t tri, bool valid) {
    RecQ r; r.tri = tri;
    const uint4* p = reinterpret_cast<const uint4*>(recs + tri);
#pragma unroll
    for (int k = 0; k < 8; ++k) r.q[k] = valid ? p[k] : make_uint4(0, 0, 0, 0);
    return r;
}
__device__ __forceinline__ uint32_t bcast_u(uint32_t v, uint32_t j) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)j); }
__device__ __forceinline__ double bcast_d(uint32_t lo, uint32_t hi, uint32_t j) {
    return __hiloint2double(__builtin_amdgcn_readlane((int)hi, (int)j), __builtin_amdgcn_readlane((int)lo, (int)j));
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

// Per-triangle constants of the pixel loop.  Every lane holds the same values: they come from ONE broadcast read of the
// triangle's slot in LDS (all lanes read the same address), which costs 8 LDS instructions and no vector-ALU issue slot;
// the 31 v_readlane that used to move them into SGPRs were 22 % of the kernel's VALU instructions (profiles/r01_raster_pmc.txt).
// Only the three words that steer control flow are made wave-uniform (v_readfirstlane).
struct TriConst {
    double ax, ay, s0x, s0y, s1x, s1y, uz, ruz, z0, z1, z2, iw0, iw1, iw2;
    double c0, g1, g2;  // depth plane for the per-pixel early test: every covered pixel has z > c0 + (ax-x) g1 + (ay-y) g2 (c0 = -inf: no test)
    uint32_t color, dl, tri;    // per-lane copies
    uint32_t rbox;      // SGPR: rx0 | ry0<<8 | (rx1-rx0)<<16 | (ry1-ry0)<<24, the clamped bbox relative to the tile origin
    uint32_t blocks;    // SGPR: bit k set: the aligned 8x8 block k of the tile may hold covered pixels that pass the z-test
    uint32_t j;         // SGPR: lane of the batch that holds this triangle's record (GOURAUD varyings ride in that lane)
};
// LDS slot of one surviving triangle of a batch: 16-byte chunks
//   0: ax ay   1: s0x s0y   2: s1x s1y   3: uz ruz   4: z0 z1   5: z2 c0   6: g1 g2   7: rbox, blocks | lane << 24, color, tri
//   (kinds other than FLAT)  8: iw0 iw1   9: iw2, dl
constexpr int TC_CHUNKS_FLAT = 8, TC_CHUNKS_ANY = 10;
#ifndef TRGL_TC_BYTES
#define TRGL_TC_BYTES 1456        // per wave: 11 FLAT slots (9 of the other kinds); with the 8.5 KB depth tile = 10 KB = 16 waves per CU
#endif
// Per-wave tile state.
#ifdef TRGL_DEBUG_COUNTERS
// counted once per wave whatever lanes are active: the first active lane takes the increment, the lanes are summed at the end
#define TRGL_DBG(i, n) do { const unsigned long long n_ = (n); if (S.lane == __ffsll((long long)__ballot(1)) - 1) S.dbg[i] += n_; } while (0)
#else
#define TRGL_DBG(i, n) ((void)0)
#endif
typedef __attribute__((address_space(3))) double lds_f64;
struct TileState {
#ifdef TRGL_DEBUG_COUNTERS
    unsigned long long dbg[16];
#endif
    int lane, px0, py0, xa1, ya0, ya1;
    double lxm, lym;        // (lane&7) + 0.5 - 2^51 and (lane>>3) + 0.5 - 2^51: pixel centre = (2^51 + block origin) + this, exactly
    uint32_t laddr;         // LDS byte address of the lane's pixel of block 0: tile base + 8 * ((lane>>3)*32 + (lane&7) + (((lane>>3)&3)<<3)), see lds_index()
    double* zt;
    uint8_t* fb_lane;       // address of this lane's pixel of block 0 in the framebuffer (colours are written straight to it)
    int bpp; uint32_t row_bytes;   // framebuffer bytes per pixel / per row
    uint32_t* id_lane; uint32_t row_px;   // PHONG / EYE flushes: this lane's pixel of block 0 in the visibility buffer, pixels per row
    uint32_t frags; double zmin, zmax; bool zero_locked;
};

// our_gl.cpp:147-199 for one triangle on one tile: 8x8 pixel blocks, one pixel per lane.
// WELL_SCALED (see k_setup) selects the division-free coverage test and the FMA division by u.z.  Three stages per block, each under
// the lanes the one before left: per-pixel depth plane, coverage, divisions + exact z-test + stores.
// KIND: the flush's shader kind when every draw has the same one (TRGL_SHADER_*), or KIND_ANY (per-triangle switch).
// For GOURAUD / PHONG / EYE the triangle's varyings sit in lane j of the batch registers V (loaded with the records,
// so the fragment branch never waits on memory) and are broadcast where a block actually shades.
constexpr int KIND_ANY = 4;
struct VaryQ { uint4 v[2]; uint32_t color; };       // GOURAUD: three intensities + the base colour of lane j's triangle

template <int KIND, bool WELL_SCALED, int BPP>
__device__ __forceinline__ void raster_triangle(const TriConst& T, const VaryQ& V, TileState& S,
                                                const DrawDesc* __restrict__ draws,
                                                const DevTexture* __restrict__ tex, DevStats* __restrict__ stats) {
    constexpr bool FLAT_ONLY = KIND == TRGL_SHADER_FLAT;
    const double uz = T.uz, ruz = T.ruz;
    // the clamped bbox relative to the tile origin (0..31), from the batch phase
    const int rx0 = (int)(T.rbox & 0xff), ry0 = (int)((T.rbox >> 8) & 0xff);
    const int rx1 = rx0 + (int)((T.rbox >> 16) & 0xff), ry1 = ry0 + (int)(T.rbox >> 24);
    // The scan walks the tile's ALIGNED 8x8 blocks named by T.blocks (bit 4*cy+cx), one pixel per lane.  The kernel
    // issues about one instruction per SIMD issue slot whatever its type (PMC: VALU + SALU + branch counts vs slots),
    // so scalar bookkeeping per block is kept as short as the vector part.
#ifdef TRGL_DEBUG_COUNTERS
    uint32_t m = 0;                                              // diagnostic build: every block of the bbox runs, and a
    for (int cy = ry0 >> 3; cy <= (ry1 >> 3); ++cy)               // block the mask dropped must not write anything
        for (int cx = rx0 >> 3; cx <= (rx1 >> 3); ++cx) m |= 1u << (4 * cy + cx);
#else
    uint32_t m = T.blocks;                                        // never 0: such triangles are not broadcast
#endif
    do {
        {
            const int k = __builtin_ctz(m);
            m &= m - 1;
            const int cx = k & 3, cy = k >> 2;
            const int k8 = (8 * cx) | (256 * cy);                             // wave-uniform: the block's offset in lds_index() units
#ifdef TRGL_DEBUG_COUNTERS
            const bool dropped = !((T.blocks >> k) & 1u);
            if (dropped) TRGL_DBG(6, 1);
#endif
            // lanes of the block inside the clamped bbox: one unsigned compare per axis (a negative difference wraps)
            const bool act = (uint32_t)((S.lane & 7) + (8 * cx - rx0)) <= (uint32_t)(rx1 - rx0) &&
                             (uint32_t)((S.lane >> 3) + (8 * cy - ry0)) <= (uint32_t)(ry1 - ry0);
            const int bx = S.px0 + 8 * cx, by = S.py0 + 8 * cy;
#ifdef TRGL_DEBUG_COUNTERS
            if (!dropped) TRGL_DBG(1, 1);                                      // blocks visited (the diagnostic build also walks the dropped ones)
#endif
            // = lds_index(x, y) for an aligned block: row and column of the lane occupy bits 5-7 and 0-2, the row swizzle and 8 cx
            // bits 3-4, 256 cy bits 8-9: the whole index is ONE xor of a per-lane constant with a scalar
            uint32_t k8b = (uint32_t)k8 << 3;
            asm("" : "+s"(k8b));                                              // one scalar, one v_xor (the compiler would split it into a three-operand xor plus a move)
            const uint32_t la = S.laddr ^ k8b;
            const double zold = *(lds_f64*)(uintptr_t)la;
            // pixel centre (x+0.5, y+0.5), our_gl.cpp:149: (2^51 + bx) + (lx + 0.5 - 2^51) is exact
            const double pxc = __hiloint2double(0x43200000, bx << 1) + S.lxm;
            const double pyc = __hiloint2double(0x43200000, by << 1) + S.lym;
            // barycentric(), our_gl.cpp:77-86 (s0.xy, s1.xy and u.z hoisted into the record)
            const double s0z = T.ax - pxc, s1z = T.ay - pyc;
            // Depth first, per pixel.  The z of :156-158 is the plane z0 + (s0z Gx + s1z Gy) / u.z through the three vertices
            // [Gx = s1x (z1-z0) - s1y (z2-z0), Gy = s0y (z2-z0) - s0x (z1-z0)] up to the roundings of u.x, u.y, the three quotients
            // and the weighted sum: at most 2^-50 max|z_i| (R S/|u.z| + 1) for a covered pixel (k_raster, block masks).  g1 = Gx/u.z and
            // g2 = Gy/u.z carry a few more roundings of the same size, and c0 = z0 minus 2^-40 max|z_i| (R S/|u.z| + 1) covers all of
            // it a thousand times over: a pixel with c0 + s0z g1 + s1z g2 >= zold fails the strict z-test of :165 whatever its
            // coverage and the low bits of its z, so only lanes that can still win run the coverage arithmetic, and a block without
            // one (47 % of the visited blocks on C4) ends here.  NaN reads as "keep"; c0 = -inf, g = 0 for a triangle that is not
            // well scaled or whose plane constants leave the normal range.
            const double zpl = __builtin_fma(s0z, T.g1, __builtin_fma(s1z, T.g2, T.c0));
#ifdef TRGL_DEBUG_COUNTERS
            const bool zkill = act && (zpl >= zold);
            const bool alive = act;
            if (!dropped) TRGL_DBG(3, __popcll(__ballot(act)));                // lanes of visited blocks inside the bbox
            if (!dropped && !__ballot(act && !zkill)) TRGL_DBG(13, 1);         // blocks the production kernel leaves at the depth-plane test
#else
            const bool alive = act && !(zpl >= zold);
#endif
            bool cov = false;
            double ux = 0.0, uy = 0.0, us = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0;
            if (alive) {
                ux = T.s0y * s1z - s0z * T.s1y;                               // geometry.h:145
                uy = s0z * T.s1x - T.s0x * s1z;                               // geometry.h:146
                us = ux + uy;
                if (WELL_SCALED) {
                    // u.z < 0 and nothing can over/underflow, so the signs of the quotients are known
                    // without dividing: u.y/u.z < 0 <=> u.y > 0, u.x/u.z < 0 <=> u.x > 0, and
                    // 1 - RN(us/u.z) < 0 <=> RN(us/u.z) > 1 <=> us < u.z  ("exactness").
                    cov = !(us < uz) && !(uy > 0.0) && !(ux > 0.0);               // :152
                } else {
                    b0 = 1.0 - us / uz;                                           // :85, as written
                    b1 = uy / uz;
                    b2 = ux / uz;
                    cov = !(b0 < 0 || b1 < 0 || b2 < 0);                          // :152
                }
            }
#ifdef TRGL_DEBUG_COUNTERS
            if (!dropped) { if (__ballot(cov)) TRGL_DBG(2, 1); else TRGL_DBG(5, 1); }   // blocks that reach the divisions / that do not
#endif
            {
                if (cov) {
                if (WELL_SCALED) {
                    b0 = 1.0 - div_by_uz(us, uz, ruz);
                    b1 = div_by_uz(uy, uz, ruz);
                    b2 = div_by_uz(ux, uz, ruz);
                }
                const double z = b0 * T.z0 + b1 * T.z1 + b2 * T.z2;           // :156-158
                // :160 — checked on both paths: the NDC depths of a well-scaled triangle are finite but not bounded
#ifdef TRGL_DEBUG_COUNTERS
                if (!dropped) {
                    const unsigned long long pass = __ballot(__builtin_isfinite(z) && (z < zold));
                    TRGL_DBG(8, __popcll(__ballot(1)));                       // lanes that run the divisions
                    TRGL_DBG(9, __popcll(pass));                              // ... and write their pixel
                    if (!pass) TRGL_DBG(10, 1);                               // blocks that ran the divisions and wrote nothing
                    if (!__ballot(!zkill)) TRGL_DBG(11, 1);                   // blocks with covered pixels that the depth-plane test ends early
                    TRGL_DBG(12, __popcll(__ballot(zkill && __builtin_isfinite(z) && (z < zold))));   // lanes it would wrongly kill (must be 0)
                }
#endif
                if (__builtin_isfinite(z) && (z < zold)) {                    // :160, :165
                    // PHONG / EYE fragments are not shaded here: the pixel remembers which triangle owns it and k_shade
                    // runs the fragment shader once per visible pixel when the list is done (the shaders have no side
                    // effects, so the image is the same as shading every z-pass in order, and the counters do not depend
                    // on colours).  FLAT / GOURAUD colours are stored at once (and, in a mixed flush, disown the pixel).
                    uint32_t color = 0;
                    bool shade_later = KIND == TRGL_SHADER_PHONG || KIND == TRGL_SHADER_EYE;      // wave-uniform
                    if (FLAT_ONLY) {
                        color = T.color;
                    } else if (KIND == TRGL_SHADER_GOURAUD || KIND == KIND_ANY) {
                        // (constant address space + wave-uniform index: the descriptor's fields come by scalar loads, see k_shade)
                        typedef const __attribute__((address_space(4))) DrawDesc CDraw;
                        CDraw& d = ((CDraw*)draws)[T.dl >> 24];
                        const int kind = KIND == KIND_ANY ? d.kind : KIND;
                        if (kind == TRGL_SHADER_FLAT) {
                            color = T.color;
                        } else if (kind == TRGL_SHADER_GOURAUD) {
                            double pc[3];
                            const double denom = b0 * T.iw0 + b1 * T.iw1 + b2 * T.iw2;            // :172-174
                            if (fabs(denom) < 1e-15) { pc[0] = b0; pc[1] = b1; pc[2] = b2; }      // :177-185
                            else { pc[0] = (b0 * T.iw0) / denom; pc[1] = (b1 * T.iw1) / denom; pc[2] = (b2 * T.iw2) / denom; }
                            if (KIND == KIND_ANY) {
                                const uint32_t local = T.dl & 0xffffffu;
                                color = frag_gouraud(d.colors ? d.colors[local] : 0xffffffffu, d.vary + (size_t)local * d.K, pc);
                            } else {
                                double vary[3];
                                vary[0] = bcast_d(V.v[0].x, V.v[0].y, T.j); vary[1] = bcast_d(V.v[0].z, V.v[0].w, T.j);
                                vary[2] = bcast_d(V.v[1].x, V.v[1].y, T.j);
                                color = frag_gouraud(bcast_u(V.color, T.j), vary, pc);
                            }
                        } else {
                            shade_later = true;
                        }
                    }
                    if (KIND >= TRGL_SHADER_PHONG && S.id_lane)               // a flush with PHONG / EYE draws (wave-uniform)
                        S.id_lane[(size_t)(8 * cy) * S.row_px + (size_t)(8 * cx)] = shade_later ? T.dl : 0xffffffffu;   // draw << 24 | triangle in its draw (draw < 64)
#ifdef TRGL_DEBUG_COUNTERS
                    if (dropped) TRGL_DBG(7, 1);                              // must stay 0
#endif
                    *(lds_f64*)(uintptr_t)la = z;                                        // :191
                    if (!shade_later) {                                       // :192, tgaimage.cpp:32-39: straight to the framebuffer
                        // the block's offset from block 0 fits 32 bits (24 rows x < 2^18 bytes): scalar arithmetic, one 64-bit add per lane
                        uint8_t* dst = S.fb_lane + (uint32_t)((uint32_t)(8 * cy) * S.row_bytes + (uint32_t)(8 * cx) * (uint32_t)S.bpp);
                        const int bpp = BPP ? BPP : S.bpp;                     // 1, 3 or 4 (trgl_create); compile-time in the FLAT kernels
                        if (bpp == 3) { *reinterpret_cast<uint16_t*>(dst) = (uint16_t)color; dst[2] = (uint8_t)(color >> 16); }   // TGAImage::set: b, g, r
                        else if (bpp == 4) *reinterpret_cast<uint32_t*>(dst) = color;
                        else dst[0] = (uint8_t)color;
                    }
                    ++S.frags;                                                // :194
                    // :197-198.  After a few fragments a lane's running min/max rarely moves, so the updates
                    // live in a branch.  A written zero can only end up as a z-range end if it is a new min or
                    // max of its lane when it is written, so the first-zero bookkeeping lives there too.
                    if ((z < S.zmin) || (S.zmax < z)) {
                        S.zmin = dmin(S.zmin, z); S.zmax = dmax(S.zmax, z);
                        if (z == 0.0 && !S.zero_locked) {
                            const int x = bx + (S.lane & 7), y = by + (S.lane >> 3);
                            unsigned long long order = ((unsigned long long)T.tri << 32) | ((unsigned long long)x << 16) | (unsigned long long)y;
                            atomicMin(__builtin_signbit(z) ? &stats->zero_neg_key : &stats->zero_pos_key, order);
                        }
                    }
                }
                }
            }
        }
    } while (m);
}

// Tile out: row-contiguous stores of the wave's rows [ya0, ya1] x columns [px0, xa1].  CLEARED: the item has no
// triangles and starts from the clear values, so they are stored directly (no LDS round trip; this is the whole
// kernel on a clear-only frame, the "framebuffer + z write-out" figure of BASELINE.json).
template <bool CLEARED>
__device__ __forceinline__ void tile_out_z(const FrameParams& fp, const double* zt, int lane,
                                         int px0, int py0, int xa1, int ya0, int ya1) {
    const bool full_x = (px0 + TRGL_TILE - 1) <= xa1;
    // z: 16 B per lane, 4 rows per store instruction
    if (full_x && (fp.W & 1) == 0) {
        for (int r4 = 0; r4 < TRGL_TILE; r4 += 4) {
            int x = px0 + ((lane & 15) << 1), y = py0 + r4 + (lane >> 4);
            if (y >= ya0 && y <= ya1) {
                double2 v = CLEARED ? make_double2(fp.clear_z, fp.clear_z) : *reinterpret_cast<const double2*>(&zt[lds_index(x, y)]);
                { typedef double nt_d2 __attribute__((ext_vector_type(2))); nt_d2 nv = {v.x, v.y}; __builtin_nontemporal_store(nv, reinterpret_cast<nt_d2*>(&fp.zb[(size_t)x + (size_t)y * fp.W])); }
            }
        }
    } else {
        for (int r2 = 0; r2 < TRGL_TILE; r2 += 2) {
            int x = px0 + (lane & 31), y = py0 + r2 + (lane >> 5);
            if (x <= xa1 && y >= ya0 && y <= ya1) fp.zb[(size_t)x + (size_t)y * fp.W] = CLEARED ? fp.clear_z : zt[lds_index(x, y)];
        }
    }
}

// The clear colour of the wave's rows x columns, row-contiguous (12 or 16 B per lane).  Ordinary stores: fragments
// overwrite these bytes later, preferably while the lines are still in L2.
__device__ __forceinline__ void tile_clear_color(const FrameParams& fp, int lane, int px0, int py0, int xa1, int ya0, int ya1) {
    const bool full_x = (px0 + TRGL_TILE - 1) <= xa1;
    // colour: 4 pixels per lane (12 B for RGB, 16 B for RGBA), 8 rows per store instruction
    if (full_x && (fp.W & 3) == 0 && (fp.bpp == 3 || fp.bpp == 4)) {
        for (int r8 = 0; r8 < TRGL_TILE; r8 += 8) {
            int x = px0 + ((lane & 7) << 2), y = py0 + r8 + (lane >> 3);
            if (y >= ya0 && y <= ya1) {
                const uint4 c = make_uint4(fp.clear_color, fp.clear_color, fp.clear_color, fp.clear_color);
                size_t idx = (size_t)x + (size_t)y * fp.W;
                if (fp.bpp == 4) {
                    *reinterpret_cast<uint4*>(fp.fb + idx * 4) = c;
                } else {
                    uint32_t d0 = (c.x & 0xffffffu) | (c.y << 24);
                    uint32_t d1 = ((c.y >> 8) & 0xffffu) | (c.z << 16);
                    uint32_t d2 = ((c.z >> 16) & 0xffu) | (c.w << 8);
                    uint32_t* dst = reinterpret_cast<uint32_t*>(fp.fb + idx * 3);
                    dst[0] = d0; dst[1] = d1; dst[2] = d2;
                }
            }
        }
    } else {
        for (int r2 = 0; r2 < TRGL_TILE; r2 += 2) {
            int x = px0 + (lane & 31), y = py0 + r2 + (lane >> 5);
            if (x <= xa1 && y >= ya0 && y <= ya1) {
                const uint32_t c = fp.clear_color;
                uint8_t* dst = fp.fb + ((size_t)x + (size_t)y * fp.W) * fp.bpp;
                for (int i = 0; i < fp.bpp; ++i) dst[i] = (uint8_t)(c >> (8 * i));
            }
        }
    }

}

// BPP: the framebuffer's bytes per pixel when the kernel is compiled for one (3 or 4, FLAT only), 0 = read from FrameParams
// ALLWS: the flush holds no triangle that needs the literal path (k_setup counted them): the kernel is compiled without it, and the
// block loop exists once (two loops sharing the wave's running statistics cost ten register moves per scanned triangle)
template <int KIND, int BPP = 0, bool ALLWS = false>
__global__ __launch_bounds__(64 * TRGL_WAVES_PER_BLOCK) __attribute__((amdgpu_waves_per_eu((KIND == TRGL_SHADER_GOURAUD || KIND == 4) ? 3 : 4, 4))) void k_raster(FrameParams fp, const TriRec* __restrict__ recs,
                                                const uint32_t* __restrict__ vals,
                                                const uint32_t* __restrict__ tile_start,
                                                const uint32_t* __restrict__ tile_end,
                                                const DrawDesc* __restrict__ draws,
                                                const DevTexture* __restrict__ tex, DevStats* __restrict__ stats,
                                                const uint32_t* __restrict__ items, const uint32_t* __restrict__ n_items,
                                                unsigned long long* __restrict__ item_stats) {
    // 8 KB per wave, first in the block's LDS and 8 KB-aligned: a block's offset inside a wave's depth tile is XORed into the lane's byte address
    __shared__ __attribute__((aligned(8192))) double s_z[TRGL_WAVES_PER_BLOCK][TRGL_TILE_PIX];
    __shared__ double s_hz[TRGL_WAVES_PER_BLOCK][64];    // depth maxima of the 64 4x4-pixel cells of the tile
    __shared__ __attribute__((aligned(16))) uint4 s_tc[TRGL_WAVES_PER_BLOCK][TRGL_TC_BYTES / 16];   // scan constants of a batch's surviving triangles

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // One wave = one work item = (tile, band of rows).  Ordinary tiles are one item; tiles whose triangle list is
    // much longer than the average are cut into 2..16 horizontal bands (k_make_items) so that a few dense tiles
    // (mesh silhouettes) do not serialise the frame: every band walks the same list and skips, after two
    // v_readlane, the triangles that miss its rows.
    const uint32_t item_idx = blockIdx.x * TRGL_WAVES_PER_BLOCK + w;
    if (item_idx >= *n_items) return;                    // no block-level barrier is used below
    const uint32_t item = items[item_idx];
    const int t = (int)(item & 0xffffffu);
    const int band = (int)((item >> 24) & 0xf), band_log2 = (int)(item >> 28);
    const int tile_y = t / fp.tiles_x, tile_x = t - tile_y * fp.tiles_x;
    const int px0 = tile_x << TRGL_TILE_LOG2, py0 = tile_y << TRGL_TILE_LOG2;
    const int band_rows = TRGL_TILE >> band_log2;
    // rows / columns of this item that exist and belong to this context's strip
    const int xa1 = min(px0 + TRGL_TILE - 1, fp.W - 1);
    const int ya0 = max(py0 + band * band_rows, fp.strip_y0);
    const int ya1 = min(min(py0 + (band + 1) * band_rows - 1, fp.H - 1), fp.strip_y1 - 1);

    uint32_t beg = tile_start[t], end = tile_end[t];
    if (!fp.init_from_clear && beg == end) {             // nothing to composite onto this tile (k_make_items skips these)
        if (lane == 0) {
            ulonglong2* dst = reinterpret_cast<ulonglong2*>(item_stats + (size_t)item_idx * 4);
            dst[0] = make_ulonglong2(0ull, ~0ull); dst[1] = make_ulonglong2(0ull, 0ull);
        }
        return;
    }

    double* zt = s_z[w];
    if (fp.init_from_clear && beg == end) {              // cleared and empty: store the clear values, nothing else
        tile_out_z<true>(fp, zt, lane, px0, py0, xa1, ya0, ya1);
        tile_clear_color(fp, lane, px0, py0, xa1, ya0, ya1);
        if (lane == 0) {
            ulonglong2* dst = reinterpret_cast<ulonglong2*>(item_stats + (size_t)item_idx * 4);
            dst[0] = make_ulonglong2(0ull, ~0ull); dst[1] = make_ulonglong2(0ull, 0ull);
        }
        return;
    }

    // ---- tile in -----------------------------------------------------------------------------------------
    // Depths live in LDS for the whole tile.  Colours do not: a fragment that passes the z-test stores its colour
    // straight into the framebuffer (TGAImage::set, tgaimage.cpp:32-39).  With aligned blocks the same lane owns a
    // pixel every time, so successive writes to a pixel are same-thread, same-address stores and keep program order.
    // 8.6 KB of LDS per wave instead of 12.6 KB = 16 waves per CU instead of 12, and no colour tile in / out.
    const bool DEFERRED = fp.idbuf != nullptr;           // the flush has PHONG / EYE draws (trgl_flush)
    if (DEFERRED) {                                      // visibility buffer of the rows this item owns: no owner yet
        for (int r2 = 0; r2 < TRGL_TILE; r2 += 2) {
            const int x = px0 + (lane & 31), y = py0 + r2 + (lane >> 5);
            if (x <= xa1 && y >= ya0 && y <= ya1) fp.idbuf[(size_t)x + (size_t)y * fp.W] = 0xffffffffu;
        }
    }
    // Pixels of the tile that this item does not own (other bands, rows outside the strip, beyond the image) hold -inf: they
    // are never scanned or stored, and the depth-maxima reduction of the batch phase needs no ownership masks.
    if (fp.init_from_clear) {
        for (int r = 0; r < TRGL_TILE; r += 2) {
            const int x = px0 + (lane & 31), y = py0 + r + (lane >> 5);
            zt[lds_index(x, y)] = (x <= xa1 && y >= ya0 && y <= ya1) ? fp.clear_z : -__builtin_inf();
        }
        tile_clear_color(fp, lane, px0, py0, xa1, ya0, ya1);
    } else {
        for (int r = 0; r < TRGL_TILE; r += 2) {
            int x = px0 + (lane & 31), y = py0 + r + (lane >> 5);
            double z = -__builtin_inf();
            if (x <= xa1 && y >= ya0 && y <= ya1) z = fp.zb[(size_t)x + (size_t)y * fp.W];
            zt[lds_index(x, y)] = z;
        }
    }
    // other lanes wrote the clear colour / empty owner of this lane's pixels: have those stores acknowledged before any
    // fragment of this wave follows them
    if (fp.init_from_clear || DEFERRED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    constexpr bool FLAT_ONLY = KIND == TRGL_SHADER_FLAT;
    // GOURAUD's 3 intensities + base colour ride along with the records (lane j of V); PHONG / EYE are shaded by k_shade.
    constexpr bool HAS_V = KIND == TRGL_SHADER_GOURAUD;
    TileState S;
#ifdef TRGL_DEBUG_COUNTERS
    for (int k = 0; k < 16; ++k) S.dbg[k] = 0;
#endif
    S.lane = lane; S.px0 = px0; S.py0 = py0; S.xa1 = xa1; S.ya0 = ya0; S.ya1 = ya1; S.zt = zt;
    S.bpp = fp.bpp; S.row_bytes = (uint32_t)fp.W * (uint32_t)fp.bpp;
    S.fb_lane = fp.fb + ((size_t)(py0 + (lane >> 3)) * fp.W + (size_t)(px0 + (lane & 7))) * fp.bpp;
    S.row_px = (uint32_t)fp.W;
    S.id_lane = DEFERRED ? fp.idbuf + ((size_t)(py0 + (lane >> 3)) * fp.W + (size_t)(px0 + (lane & 7))) : nullptr;
    S.lxm = ((double)(lane & 7) + 0.5) - 0x1p51; S.lym = ((double)(lane >> 3) + 0.5) - 0x1p51;
    S.laddr = (uint32_t)(uintptr_t)(lds_f64*)zt + 8u * (uint32_t)(((lane >> 3) * 32 + (lane & 7)) | (((lane >> 3) & 3) << 3));
    S.frags = 0; S.zmin = __builtin_inf(); S.zmax = -__builtin_inf();
    S.zero_locked = stats->zero_locked != 0;

    // ---- the tile's triangles, in submission order --------------------------------------------
    // 64 records at a time: lane i holds the 128-B record of the batch's i-th triangle in registers for the batch phase
    // (block masks, one triangle per lane); the survivors' constants then go through LDS slots to all lanes.
    RecQ cur;
    uint32_t next_tri = 0;               // list entry of this lane in the NEXT batch: fetched one batch ahead, so that the records
                                         // of a batch are one memory latency away when its turn comes, not two
    if (beg < end) {                     // an empty tile must not touch vals/recs at all
        uint32_t p = beg + lane;
        cur = load_rec(recs, vals[p < end ? p : end - 1], true);
        if (beg + 64 < end) { p += 64; next_tri = vals[p < end ? p : end - 1]; }
    } else {
        cur.tri = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) cur.q[k] = make_uint4(0, 0, 0, 0);
    }
    double* hz = s_hz[w];
    uint4* tc = s_tc[w];
    VaryQ V;
    V.color = 0; V.v[0] = make_uint4(0, 0, 0, 0); V.v[1] = make_uint4(0, 0, 0, 0);
    for (uint32_t bs = beg; bs < end; bs += 64) {
        const uint32_t nbatch = min(64u, end - bs);
        // ---- block masks, once per batch of 64 list entries ----------------------------------------------------
        // Lane l owns the 4x4-pixel cell (l&7, l>>3) of the tile and reduces the depths currently stored in it (rows
        // and columns this item does not own count as -inf: they are never scanned); 16 lanes then publish the
        // maxima of the 16 aligned 8x8 blocks.  Cell maxima only fall while a batch is rasterized, so the values
        // of the batch start stay valid upper bounds.
        //
        // Then every lane decides, for ITS OWN triangle of the batch, which aligned 8x8 blocks of the tile have to be
        // scanned at all.  Block k is dropped when
        //  (edges)  one of the three edge functions excludes all of it: a pixel is covered iff the rounded u.x <= 0,
        //           u.y <= 0 and u.x+u.y >= u.z (raster_triangle); each is affine in the pixel centre, so its extreme
        //           over the block sits at the corner the gradient's signs select;
        //  (depth)  no covered pixel of it can pass the strict z-test (our_gl.cpp:165): every covered pixel has
        //           b_i >= 0 and b0+b1+b2 = 1 +- 2^-50, hence z >= zbound = min(z0,z1,z2) - 2^-40 max|z_i|; and the
        //           computed z (our_gl.cpp:156-158) differs from the depth plane z0 + (u.y/u.z)(z1-z0) + (u.x/u.z)(z2-z0)
        //           by at most 2^-50 max|z_i| (R S/|u.z| + 1)  [R >= |A - pixel| on the tile, S = sum of |edge deltas|:
        //           the roundings of u.x, u.y, their quotients and the weighted sum], so z >= plane minimum over the
        //           block - that; if the larger of the two bounds is >= the block's stored maximum, nothing passes.
        // Corner values are stepped block to block; those roundings and a mis-chosen corner of a nearly flat function
        // are of the same 2^-50 order, and every test keeps a 2^-40 margin.  NaN/inf compare false = "keep".
        // A triangle with no block left is skipped before its constants are broadcast.  On C4 this leaves 0.5x list
        // entries and 0.6x blocks of what a bbox scan with a per-triangle depth bound visits
        // (profiles/raster_counters.py; the diagnostic build checks that no dropped block would have written).
        float bt;                                  // lane 16 r + 2 c: float upper bound of the stored maximum of the aligned 8x8 block (c, r)
        {
            const int cx = lane & 7, cy = lane >> 3;
            double m = -__builtin_inf();
#pragma unroll
            for (int dy = 0; dy < 4; ++dy) {
#pragma unroll
                for (int dx = 0; dx < 4; ++dx) m = vmax(m, zt[lds_index(4 * cx + dx, 4 * cy + dy)]);
            }
            hz[lane] = m;
            __builtin_amdgcn_wave_barrier();
            const int f = (cy & 6) * 8 + (cx & 6);                                  // first cell of this lane's block
            bt = f32_up(vmax(vmax(hz[f], hz[f + 1]), vmax(hz[f + 8], hz[f + 9])));
            // bt is read with v_readlane from OTHER lanes inside the per-lane branch below: pin it here, in uniform control flow,
            // or the compiler sinks part of its computation into the branch and the inactive lanes hold garbage
            asm volatile("" : "+v"(bt));
            __builtin_amdgcn_wave_barrier();
        }
        unsigned long long todo;
        uint32_t blocks_l = 0;                    // ... and the aligned 8x8 blocks of the tile its scan has to visit
        uint32_t rbox_l = 0;                      // ... and its clamped bbox relative to the tile origin (TriConst::rbox)
        {
            const int bx0 = (int)(cur.q[7].x & 0xffff), by0 = (int)(cur.q[7].x >> 16);
            const int bx1 = (int)(cur.q[7].y & 0xffff), by1 = (int)(cur.q[7].y >> 16);
            const int x0 = max(bx0, px0), x1 = min(bx1, xa1), y0 = max(by0, ya0), y1 = min(by1, ya1);
            bool skip = (uint32_t)lane >= nbatch || x0 > x1 || y0 > y1;      // not in the list / misses this band
            const double ruz_l = __hiloint2double((int)cur.q[3].w, (int)cur.q[3].z);
            if (!skip) {
                rbox_l = (uint32_t)(x0 - px0) | ((uint32_t)(y0 - py0) << 8) | ((uint32_t)(x1 - x0) << 16) | ((uint32_t)(y1 - y0) << 24);
                const int c0 = (x0 - px0) >> 3, c1 = (x1 - px0) >> 3, r0 = (y0 - py0) >> 3, r1 = (y1 - py0) >> 3;
                if (ruz_l == 0.0) {                // not well scaled: the literal path scans every block of the bbox
                    blocks_l = (((2u << c1) - (1u << c0)) & 0xfu) * 0x1111u & ((0xffffu >> (12 - 4 * r1)) & (0xffffu << (4 * r0)));
                } else {
                    const double z0 = __hiloint2double((int)cur.q[4].y, (int)cur.q[4].x);
                    const double z1 = __hiloint2double((int)cur.q[4].w, (int)cur.q[4].z);
                    const double z2 = __hiloint2double((int)cur.q[5].y, (int)cur.q[5].x);
                    const double zlo = dmin(dmin(z0, z1), z2), zabs = dmax(dmax(fabs(z0), fabs(z1)), fabs(z2));
                    const double zbound = zlo - zabs * 0x1p-40;
                    const double ax = __hiloint2double((int)cur.q[0].y, (int)cur.q[0].x), ay = __hiloint2double((int)cur.q[0].w, (int)cur.q[0].z);
                    const double s0x = __hiloint2double((int)cur.q[1].y, (int)cur.q[1].x), s0y = __hiloint2double((int)cur.q[1].w, (int)cur.q[1].z);
                    const double s1x = __hiloint2double((int)cur.q[2].y, (int)cur.q[2].x), s1y = __hiloint2double((int)cur.q[2].w, (int)cur.q[2].z);
                    const double uz_l = __hiloint2double((int)cur.q[3].y, (int)cur.q[3].x);
                    // ---- the 16 block tests in SINGLE precision, straight-line ----
                    // They only have to be conservative, not exact: every quantity is evaluated in float from the tile's block
                    // (0,0) and stepped by FMAs, and the margins grow from 2^-40 to 2^-18 of the same magnitudes (S R for the edge
                    // functions, zabs (R S / |u.z| + 1) for the depth plane), which covers the float roundings (a few 2^-24 of
                    // those magnitudes: conversions of the inputs, products, sums, steps, a corner mis-chosen for a nearly flat
                    // function) with a factor of >= 16 to spare and still is 1e-4 of a pixel.  Bounds that must not move the
                    // wrong way are rounded outwards (u.z and zbound down, the stored maxima up); an absolute 2^-100 keeps the
                    // margins above float denormal noise; overflow gives inf / NaN, which compare as "keep".
                    // No loop whose trip count is the largest bbox of the batch, no LDS reads, half the issue cycles of fp64.
                    const float X0 = (float)px0 + 0.5f, Y0 = (float)py0 + 0.5f;          // exact: px0 < 2^16
                    const float dx0 = (float)(ax - (double)X0), dy0 = (float)(ay - (double)Y0);
                    const float fs0x = (float)s0x, fs0y = (float)s0y, fs1x = (float)s1x, fs1y = (float)s1y, fruz = (float)ruz_l;
                    const float fuz = f32_down(uz_l), fzb = f32_down(zbound);
                    // edge functions: u.x = s0y (ay-Y) - (ax-X) s1y, u.y = (ax-X) s1x - s0x (ay-Y), u.x + u.y
                    const float gx = fs1y - fs1x, gy = fs0x - fs0y;                          // gradient of u.x + u.y
                    const float oxa = fs1y >= 0.f ? 0.f : 7.f, oya = fs0y <= 0.f ? 0.f : 7.f;    // corner of min u.x
                    const float oxb = fs1x <= 0.f ? 0.f : 7.f, oyb = fs0x >= 0.f ? 0.f : 7.f;    // corner of min u.y
                    const float oxc = gx >= 0.f ? 7.f : 0.f, oyc = gy >= 0.f ? 7.f : 0.f;        // corner of max u.x+u.y
                    const float fa0 = fs0y * (dy0 - oya) - (dx0 - oxa) * fs1y;
                    const float fb0 = (dx0 - oxb) * fs1x - fs0x * (dy0 - oyb);
                    const float fc0 = (fs0y * (dy0 - oyc) - (dx0 - oxc) * fs1y) + ((dx0 - oxc) * fs1x - fs0x * (dy0 - oyc));
                    const float R = fabsf(dx0) + fabsf(dy0) + 64.0f;                      // >= |A - pixel| (L1) for every pixel of the tile
                    const float Sa = fabsf(fs0y) + fabsf(fs1y), Sb = fabsf(fs0x) + fabsf(fs1x);
                    const float ma = 0x1p-18f * (Sa * R) + 0x1p-100f, mb = 0x1p-18f * (Sb * R) + 0x1p-100f;
                    const float lim_c = fuz - (ma + mb);
                    // depth plane: z0 + (u.y/u.z) dz1 + (u.x/u.z) dz2, minimum corner by the gradient's signs
                    const float fz0v = (float)z0, dz1 = (float)(z1 - z0), dz2 = (float)(z2 - z0), fzabs = (float)zabs;
                    const float gzx = (fs1y * dz2 - fs1x * dz1) * fruz, gzy = (fs0x * dz1 - fs0y * dz2) * fruz;
                    const float ux0 = fs0y * dy0 - dx0 * fs1y, uy0 = dx0 * fs1x - fs0x * dy0;
                    const float mz = 0x1p-18f * (fzabs * ((R * (Sa + Sb)) * fabsf(fruz) + 1.0f)) + 0x1p-100f;
                    const float fz0 = ((fz0v + (uy0 * fruz) * dz1) + (ux0 * fruz) * dz2) + ((gzx >= 0.f ? 0.f : 7.f) * gzx + (gzy >= 0.f ? 0.f : 7.f) * gzy) - mz;
                    const float sax = 8.f * fs1y, say = -8.f * fs0y, sbx = -8.f * fs1x, sby = 8.f * fs0x, scx = 8.f * gx, scy = 8.f * gy;
                    const float szx = 8.f * gzx, szy = 8.f * gzy;
                    // One number per block: keep <=> min(ma - fa, mb - fb, fc - lim_c, top - max(fz, fzb)) > 0.  (Dropping at
                    // equality with a margin is still conservative.)  v_min / v_max skip a NaN operand, a NaN result reads as
                    // "keep" through the integer compare, and triangles whose magnitudes could overflow float keep every block.
                    uint32_t mk = 0;
                    const float dma = ma - fa0, dmb = mb - fb0, dmc = fc0 - lim_c;          // slacks at block (0,0); they step linearly
                    float da_r = dma, db_r = dmb, dc_r = dmc, fz_r = fz0;
#pragma nounroll
                    for (int r = 0; r < 4; ++r) {                 // a real loop (wave-uniform trip count 4): keeps the register pressure of one row
                        uint32_t row = 0;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float da = __builtin_fmaf(-(float)c, sax, da_r), db = __builtin_fmaf(-(float)c, sbx, db_r);
                            const float dc = __builtin_fmaf((float)c, scx, dc_r), fz = __builtin_fmaf((float)c, szx, fz_r);
                            const float top = __uint_as_float(bcast_u(__float_as_uint(bt), (uint32_t)(16 * r + 2 * c)));
                            const float slack = fmin3(fmin3(da, db, dc), top - fmax2(fz, fzb), top - fmax2(fz, fzb));
                            row |= ((int)__float_as_uint(slack) > 0) ? (1u << c) : 0u;
                        }
                        mk |= row << (4 * r);
                        da_r -= say; db_r -= sby; dc_r += scy; fz_r += szy;
                    }
                    if (!(R * (Sa + Sb) < 0x1p100f)) mk = 0xffffu;       // magnitudes float cannot hold (or NaN): no block test is trusted
                    // only the blocks the clamped bbox reaches
                    mk &= (((2u << c1) - (1u << c0)) & 0xfu) * 0x1111u & ((0xffffu >> (12 - 4 * r1)) & (0xffffu << (4 * r0)));
                    blocks_l = mk;
                }
                skip = blocks_l == 0;
                if (ruz_l != 0.0) blocks_l |= 0x10000u;       // bit 16: well scaled (selects the division-free scan)
            }
            todo = __ballot(!skip);
            if (HAS_V && !skip) {        // this lane's triangle will be rasterized: fetch its varyings now
                const uint32_t dl = cur.q[7].w;
                const DrawDesc& d = draws[dl >> 24];
                const uint32_t local = dl & 0xffffffu;
                const double* vp = d.vary + (size_t)local * 3;
                const double a = vp[0], b = vp[1], c2 = vp[2];
                V.v[0] = make_uint4((uint32_t)__double2loint(a), (uint32_t)__double2hiint(a), (uint32_t)__double2loint(b), (uint32_t)__double2hiint(b));
                V.v[1] = make_uint4((uint32_t)__double2loint(c2), (uint32_t)__double2hiint(c2), 0u, 0u);
                V.color = d.colors ? d.colors[local] : 0xffffffffu;
            }
        }
        // ---- the surviving triangles, in list order (= ascending lane): constants to LDS, then one broadcast read each ----
        // The survivors of the batch (57 % of its entries on C4) write their scan constants into consecutive 128-byte slots
        // of the wave's LDS area; a slot is read back by ALL lanes at the same address.  When there are more survivors than
        // slots the batch is served in rounds.
        constexpr int CH = FLAT_ONLY ? TC_CHUNKS_FLAT : TC_CHUNKS_ANY;
        constexpr uint32_t SLOTS = TRGL_TC_BYTES / (16 * CH);
        const unsigned long long lanes_below = (1ull << lane) - 1ull;
        while (todo) {
            const uint32_t rank = (uint32_t)__popcll(todo & lanes_below);
            const bool mine = ((todo >> lane) & 1ull) && rank < SLOTS;
            if (mine) {
                uint4* d = tc + rank * CH;
                d[0] = cur.q[0]; d[1] = cur.q[1]; d[2] = cur.q[2]; d[3] = cur.q[3]; d[4] = cur.q[4];
                {   // depth plane of the per-pixel early test (raster_triangle): c0 + (ax-x) g1 + (ay-y) g2 < every covered pixel's z
                    double c0 = -__builtin_inf(), g1 = 0.0, g2 = 0.0;
                    if (blocks_l & 0x10000u) {
                        const double ax = __hiloint2double((int)cur.q[0].y, (int)cur.q[0].x), ay = __hiloint2double((int)cur.q[0].w, (int)cur.q[0].z);
                        const double s0x = __hiloint2double((int)cur.q[1].y, (int)cur.q[1].x), s0y = __hiloint2double((int)cur.q[1].w, (int)cur.q[1].z);
                        const double s1x = __hiloint2double((int)cur.q[2].y, (int)cur.q[2].x), s1y = __hiloint2double((int)cur.q[2].w, (int)cur.q[2].z);
                        const double ruz = __hiloint2double((int)cur.q[3].w, (int)cur.q[3].z);
                        const double z0 = __hiloint2double((int)cur.q[4].y, (int)cur.q[4].x);
                        const double z1 = __hiloint2double((int)cur.q[4].w, (int)cur.q[4].z);
                        const double z2 = __hiloint2double((int)cur.q[5].y, (int)cur.q[5].x);
                        const double zabs = dmax(dmax(fabs(z0), fabs(z1)), fabs(z2));
                        const double dz1 = z1 - z0, dz2 = z2 - z0;
                        const double h1 = (s1x * dz1 - s1y * dz2) * ruz, h2 = (s0y * dz2 - s0x * dz1) * ruz;
                        const double R = fabs(ax - ((double)px0 + 0.5)) + fabs(ay - ((double)py0 + 0.5)) + 64.0;   // >= |A - pixel| (L1) on the tile
                        const double Ssum = (fabs(s0x) + fabs(s0y)) + (fabs(s1x) + fabs(s1y));
                        const double mz = zabs * 0x1p-40 * (R * Ssum * fabs(ruz) + 1.0) + 0x1p-600;
                        // trusted only while nothing can overflow (R |g| bounds each product of the test); NaN compares false
                        const bool ok = zabs < 0x1p1000 && R * fabs(h1) < 0x1p900 && R * fabs(h2) < 0x1p900 && mz < 0x1p1000;
                        if (ok) { c0 = z0 - mz; g1 = h1; g2 = h2; }
                    }
                    d[5] = make_uint4(cur.q[5].x, cur.q[5].y, (uint32_t)__double2loint(c0), (uint32_t)__double2hiint(c0));
                    d[6] = make_uint4((uint32_t)__double2loint(g1), (uint32_t)__double2hiint(g1), (uint32_t)__double2loint(g2), (uint32_t)__double2hiint(g2));
                }
                d[7] = make_uint4(rbox_l, blocks_l | ((uint32_t)lane << 24), cur.q[7].z, cur.tri);
                if (!FLAT_ONLY) {
                    d[8] = make_uint4(cur.q[5].z, cur.q[5].w, cur.q[6].x, cur.q[6].y);
                    d[9] = make_uint4(cur.q[6].z, cur.q[6].w, cur.q[7].w, 0u);
                }
            }
            const unsigned long long round = __ballot(mine);
            todo &= ~round;
            const uint32_t n_round = (uint32_t)__popcll(round);
            __builtin_amdgcn_wave_barrier();
            for (uint32_t sl = 0; sl < n_round; ++sl) {
                const uint4* p = tc + sl * CH;                                  // the same address in every lane
                const double2* pd = reinterpret_cast<const double2*>(p);
                const double2 d0 = pd[0], d1 = pd[1], d2 = pd[2], d3 = pd[3], d4 = pd[4], d5 = pd[5], d6 = pd[6];
                const uint4 c6 = p[7];
                TriConst T;
                T.ax = d0.x; T.ay = d0.y; T.s0x = d1.x; T.s0y = d1.y; T.s1x = d2.x; T.s1y = d2.y;
                T.uz = d3.x; T.ruz = d3.y; T.z0 = d4.x; T.z1 = d4.y; T.z2 = d5.x; T.c0 = d5.y; T.g1 = d6.x; T.g2 = d6.y;
                T.rbox = (uint32_t)__builtin_amdgcn_readfirstlane((int)c6.x);
                const uint32_t bw = (uint32_t)__builtin_amdgcn_readfirstlane((int)c6.y);
                T.blocks = bw & 0xffffu; T.j = bw >> 24;
                T.color = c6.z; T.tri = c6.w;
                T.iw0 = T.iw1 = T.iw2 = 0.0; T.dl = 0;
                if (!FLAT_ONLY) {
                    const double2 d8 = pd[8];
                    const uint4 c9 = p[9];
                    T.iw0 = d8.x; T.iw1 = d8.y; T.iw2 = __hiloint2double((int)c9.y, (int)c9.x);
                    T.dl = (uint32_t)__builtin_amdgcn_readfirstlane((int)c9.z);
                }
                TRGL_DBG(0, 1);                                                    // list entries rasterized (not skipped)
                // "well scaled" is a property of the triangle (k_setup leaves ruz = 0 otherwise): wave-uniform by construction
                if (ALLWS || (bw & 0x10000u)) raster_triangle<KIND, true, BPP>(T, V, S, draws, tex, stats);
                else raster_triangle<KIND, false, BPP>(T, V, S, draws, tex, stats);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (bs + 64 < end) {              // (holding the next batch's records during the scan costs a wave per SIMD; measured, not kept)
            cur = load_rec(recs, next_tri, true);
            if (bs + 128 < end) { const uint32_t p = bs + 128 + lane; next_tri = vals[p < end ? p : end - 1]; }
        }
    }

    uint32_t frags = S.frags;
    const double zmin = S.zmin, zmax = S.zmax;
#ifdef TRGL_DEBUG_COUNTERS
    for (int k = 0; k < 16; ++k) if (k != 4 && S.dbg[k]) atomicAdd(&stats->dbg[k], S.dbg[k]);
    if (lane == 0) atomicAdd(&stats->dbg[4], (unsigned long long)(end - beg));
#endif

    // ---- tile out ----------------------------------------------------------------------------
    tile_out_z<false>(fp, zt, lane, px0, py0, xa1, ya0, ya1);

    // ---- stats: our_gl.cpp:194-198, reduced per wave, one set of atomics per tile -----------------
    unsigned long long kmin = zkey(zmin), kmax = zkey(zmax);
    for (int o = 32; o; o >>= 1) {
        frags += __shfl_xor(frags, o);
        unsigned long long a = __shfl_xor(kmin, o); kmin = a < kmin ? a : kmin;
        unsigned long long b = __shfl_xor(kmax, o); kmax = b > kmax ? b : kmax;
    }
    // one partial per work item, reduced by k_fold_stats: same-address atomics from thousands of items would
    // serialise at ~11 ns each
    if (lane == 0) {
        ulonglong2* dst = reinterpret_cast<ulonglong2*>(item_stats + (size_t)item_idx * 4);
        dst[0] = make_ulonglong2((unsigned long long)frags, kmin);
        dst[1] = make_ulonglong2(kmax, 0ull);
    }
}


// ---------------------------------------------------------------------------------------------
// k_shade<PHONG|EYE>: the fragment stage of a PHONG / EYE flush, once per visible pixel.
// k_raster left, for every pixel whose depth it wrote in this flush, the record index of the LAST triangle that passed the
// z-test there (fp.idbuf).  IShader::fragment has no side effects and always returns discard = false (main.cpp:92-170,
// 220-261), so calling it for that triangle only gives the same framebuffer as calling it for every z-pass in order
// (our_gl.cpp:187-192) - with 64 busy lanes per wave instead of the few pixels of one small triangle.  One 256-thread
// block per work item of k_raster (so tiles this flush did not touch are not visited), a row of 4 aligned 8x8 blocks per wave; the
// barycentrics are recomputed per pixel with exactly the operations of the scan (same bits).
// ---------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_shade(FrameParams fp, const TriRec* __restrict__ recs, const DrawDesc* __restrict__ draws,
                                                const DevTexture* __restrict__ tex, const uint32_t* __restrict__ tile_start,
                                                const uint32_t* __restrict__ tile_end, const uint32_t* __restrict__ items,
                                                const uint32_t* __restrict__ n_items) {
    const int lane = threadIdx.x & 63;
    const uint32_t item_idx = blockIdx.x;                // one 256-thread block per work item: wave w shades block row w
    if (item_idx >= *n_items) return;
    const uint32_t item = items[item_idx];
    const int t = (int)(item & 0xffffffu);
    if (tile_start[t] == tile_end[t]) return;            // no triangles: k_raster did not touch this tile's owners
    const int band = (int)((item >> 24) & 0xf), band_log2 = (int)(item >> 28);
    const int tile_y = t / fp.tiles_x, tile_x = t - tile_y * fp.tiles_x;
    const int px0 = tile_x << TRGL_TILE_LOG2, py0 = tile_y << TRGL_TILE_LOG2;
    const int band_rows = TRGL_TILE >> band_log2;
    const int xa1 = min(px0 + TRGL_TILE - 1, fp.W - 1);
    const int ya0 = max(py0 + band * band_rows, fp.strip_y0);
    const int ya1 = min(min(py0 + (band + 1) * band_rows - 1, fp.H - 1), fp.strip_y1 - 1);
    const int krow = 4 * (int)(threadIdx.x >> 6);
    // The kernel is bound by dependent memory latencies (68 % of its wave time sat in s_waitcnt): owner -> record -> varyings -> texels
    // for each of the wave's four blocks, sixteen in a row.  The four owners are fetched up front, and the visibility buffer holds
    // `draw << 24 | triangle in its draw`, from which BOTH the record (recs[draw.first + triangle]) and the varyings are addressed:
    // two loads in flight instead of one after the other.
    uint32_t owner[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = krow + j;
        const int x = px0 + 8 * (k & 3) + (lane & 7), y = py0 + 8 * (k >> 2) + (lane >> 3);
        const bool mine = x <= xa1 && y >= ya0 && y <= ya1;
        owner[j] = mine ? fp.idbuf[(size_t)x + (size_t)y * fp.W] : 0xffffffffu;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = krow + j;
        const int x = px0 + 8 * (k & 3) + (lane & 7), y = py0 + 8 * (k >> 2) + (lane >> 3);
        const size_t idx = (size_t)x + (size_t)y * fp.W;
        const uint32_t dl = owner[j];
        if (dl == 0xffffffffu) continue;
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
                const double* vary = d.vary + (size_t)local * 24;
                // barycentric(), our_gl.cpp:77-86, as in raster_triangle
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
                const double denom = b0 * r.iw0 + b1 * r.iw1 + b2 * r.iw2;                        // our_gl.cpp:172-174
                if (fabs(denom) < 1e-15) { pc[0] = b0; pc[1] = b1; pc[2] = b2; }                  // :177-185
                else { pc[0] = (b0 * r.iw0) / denom; pc[1] = (b1 * r.iw1) / denom; pc[2] = (b2 * r.iw2) / denom; }
                const int kind = KIND == KIND_ANY ? d.kind : KIND;        // wave-uniform inside this iteration
                color = kind == TRGL_SHADER_PHONG ? frag_phong(d.u, ctex, vary, pc).bgra : frag_eye(d.u, ctex, vary, pc).bgra;
            }
        }
        uint8_t* dst = fp.fb + idx * fp.bpp;                                               // TGAImage::set, tgaimage.cpp:32-39
        if (fp.bpp == 3) { dst[0] = (uint8_t)color; dst[1] = (uint8_t)(color >> 8); dst[2] = (uint8_t)(color >> 16); }
        else if (fp.bpp == 4) *reinterpret_cast<uint32_t*>(dst) = color;
        else dst[0] = (uint8_t)color;                                  // bpp is 1, 3 or 4 (trgl_create)
    }
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

// Work items of the raster kernel: one per tile row-band (see k_raster).  A tile is cut into bands when its list
// is longer than `split_len` (chosen by the host relative to the mean list length).  Tiles outside the strip's
// tile rows, and empty tiles of a flush that does not start from clear, get no item.
__global__ __launch_bounds__(256) void k_make_items(FrameParams fp, const uint32_t* __restrict__ tile_start,
                                                    const uint32_t* __restrict__ tile_end, uint32_t split_len,
                                                    uint32_t* __restrict__ items, uint32_t* __restrict__ n_items) {
    const int ntiles_strip = (fp.strip_ty1 - fp.strip_ty0) * fp.tiles_x;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t nb = 0, bl = 0, t = 0;
    if (k < ntiles_strip) {
        t = (uint32_t)(fp.strip_ty0 * fp.tiles_x + k);
        const uint32_t n = tile_end[t] - tile_start[t];
        if ((n || fp.init_from_clear) && tile_row_owned(fp, (int)(t / (uint32_t)fp.tiles_x))) {
            while (bl < 4 && (n >> bl) > split_len) ++bl;      // 1, 2, 4, 8 or 16 bands
            nb = 1u << bl;
        }
    }
    // wave-aggregated append (order of items is irrelevant)
    uint32_t inc = nb;
    const int lane = threadIdx.x & 63;
    {   // tiles with triangles, for the host's band heuristic of the NEXT flush (n_items[1])
        const unsigned long long ne = __ballot(nb != 0 && tile_end[t] != tile_start[t]);
        if (ne && lane == 0) atomicAdd(n_items + 1, (uint32_t)__popcll(ne));
    }
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if (lane >= o) inc += v; }
    const uint32_t total = __shfl(inc, 63);
    uint32_t base = 0;
    if (lane == 63 && total) base = atomicAdd(n_items, total);
    base = __shfl(base, 63) + inc - nb;
    for (uint32_t b = 0; b < nb; ++b) items[base + b] = t | (b << 24) | (bl << 28);
}

// after the raster kernel of a flush: fold the per-item partials into the context's counters (our_gl.cpp:194-198)
// and fix the sign of a zero z-range end (see DevStats).  One block.
__global__ __launch_bounds__(1024) void k_fold_stats(DevStats* __restrict__ s, uint32_t* __restrict__ n_items,
                                                     const unsigned long long* __restrict__ item_stats) {
    __shared__ unsigned long long sh[3][16];
    const uint32_t n = *n_items;
    unsigned long long fr = 0, kmin = ~0ull, kmax = 0ull;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
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
        s->fragments += fr;
        if (fr) {       // items without fragments carry the neutral keys
            if (kmin < s->zmin_key) s->zmin_key = kmin;
            if (kmax > s->zmax_key) s->zmax_key = kmax;
        }
        if (!s->zero_locked && (s->zero_pos_key != TRGL_ZERO_KEY_EMPTY || s->zero_neg_key != TRGL_ZERO_KEY_EMPTY)) {
            s->zero_sign = s->zero_neg_key < s->zero_pos_key ? 1u : 0u;
            s->zero_locked = 1u;
        }
        s->literal_tris = 0;                       // counted per flush (k_setup)
        s->zero_pos_key = TRGL_ZERO_KEY_EMPTY;
        s->zero_neg_key = TRGL_ZERO_KEY_EMPTY;
        *n_items = 0;           // every thread read it before the barrier; k_make_items of the next flush appends from 0
        s->nonempty_tiles = n_items[1]; n_items[1] = 0;
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

uint32_t raster_max_items(const FrameParams& fp, uint64_t pairs, uint32_t split_len) {
    const uint64_t tiles = owned_tiles(fp);
    // a tile with n entries makes at most max(1, 2n/split_len) bands (<= 16)
    uint64_t extra = split_len ? 2 * pairs / split_len : 0;
    if (extra > tiles * 15) extra = tiles * 15;
    return (uint32_t)(tiles + extra);
}

void launch_raster(hipStream_t s, const FrameParams& fp, int kind /* TRGL_SHADER_* if uniform over the flush, else -1 */, bool all_well_scaled, const TriRec* recs, const uint32_t* vals,
                   const uint32_t* tile_start, const uint32_t* tile_end, const DrawDesc* draws,
                   const DevTexture* tex, DevStats* stats, uint32_t split_len, uint32_t max_items, uint32_t* items,
                   uint32_t* n_items, unsigned long long* item_stats, hipEvent_t ev_before, hipEvent_t ev_after) {
    const int tiles = (fp.strip_ty1 - fp.strip_ty0) * fp.tiles_x;
    if (tiles <= 0) {
        if (ev_before) (void)hipEventRecord(ev_before, s);
        if (ev_after) (void)hipEventRecord(ev_after, s);
        return;
    }
    hipLaunchKernelGGL(k_make_items, dim3((tiles + 255) / 256), dim3(256), 0, s, fp, tile_start, tile_end, split_len, items, n_items);
    dim3 grid((max_items + TRGL_WAVES_PER_BLOCK - 1) / TRGL_WAVES_PER_BLOCK);
    if (ev_before) (void)hipEventRecord(ev_before, s);
#define TRGL_LAUNCH_RASTER(...) hipLaunchKernelGGL((k_raster<__VA_ARGS__>), grid, dim3(64 * TRGL_WAVES_PER_BLOCK), 0, s, fp, recs, vals, tile_start, tile_end, draws, tex, stats, items, n_items, item_stats)
    switch (kind) {
    case TRGL_SHADER_FLAT:
        if (all_well_scaled) {
            if (fp.bpp == 3) TRGL_LAUNCH_RASTER(TRGL_SHADER_FLAT, 3, true);
            else if (fp.bpp == 4) TRGL_LAUNCH_RASTER(TRGL_SHADER_FLAT, 4, true);
            else TRGL_LAUNCH_RASTER(TRGL_SHADER_FLAT, 0, true);
        } else {
            if (fp.bpp == 3) TRGL_LAUNCH_RASTER(TRGL_SHADER_FLAT, 3);
            else if (fp.bpp == 4) TRGL_LAUNCH_RASTER(TRGL_SHADER_FLAT, 4);
            else TRGL_LAUNCH_RASTER(TRGL_SHADER_FLAT);
        }
        break;
    case TRGL_SHADER_GOURAUD: TRGL_LAUNCH_RASTER(TRGL_SHADER_GOURAUD); break;
    case TRGL_SHADER_PHONG:   if (all_well_scaled) TRGL_LAUNCH_RASTER(TRGL_SHADER_PHONG, 0, true); else TRGL_LAUNCH_RASTER(TRGL_SHADER_PHONG); break;
    case TRGL_SHADER_EYE:     if (all_well_scaled) TRGL_LAUNCH_RASTER(TRGL_SHADER_EYE, 0, true); else TRGL_LAUNCH_RASTER(TRGL_SHADER_EYE); break;
    default:                  TRGL_LAUNCH_RASTER(KIND_ANY); break;
    }
#undef TRGL_LAUNCH_RASTER
    if (ev_after) (void)hipEventRecord(ev_after, s);
    if (fp.idbuf) {                                     // the flush has PHONG / EYE draws: shade the visible pixels they own
#define TRGL_LAUNCH_SHADE(K) hipLaunchKernelGGL(k_shade<K>, dim3(max_items), dim3(256), 0, s, fp, recs, draws, tex, tile_start, tile_end, items, n_items)
        if (kind == TRGL_SHADER_PHONG) TRGL_LAUNCH_SHADE(TRGL_SHADER_PHONG);
        else if (kind == TRGL_SHADER_EYE) TRGL_LAUNCH_SHADE(TRGL_SHADER_EYE);
        else TRGL_LAUNCH_SHADE(KIND_ANY);
#undef TRGL_LAUNCH_SHADE
    }
    hipLaunchKernelGGL(k_fold_stats, dim3(1), dim3(1024), 0, s, stats, n_items, item_stats);
}

}  // namespace trgl
