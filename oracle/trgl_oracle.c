/*
 * trgl_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT.  See trgl_oracle.h for scope and pinning.
 *
 * Every function names the reference lines it follows.  The operation ORDER of the reference is
 * kept (each product and sum rounded separately: build with -ffp-contract=off), because coverage
 * (`>= 0` on quotients), the z-test winner and the written z bits depend on the low bits.
 */
#include "trgl_oracle.h"

#include <limits.h>
#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846   /* our_gl.h:10-12 */
#endif
#include <string.h>

/* (int)double as the reference's x86-64 build executes it (cvttsd2si): values that do not fit
 * give INT_MIN ("integer indefinite").  Written out so the oracle does not rest on C UB.
 * Used wherever the reference casts: our_gl.cpp:130-133, model.cpp:420-423,433-436,452-455. */
static int x86_cvttsd2si(double d) {
    if (!(d > -2147483649.0 && d < 2147483648.0)) return INT_MIN;
    return (int)d;
}

static double dmax(double a, double b) { return (a < b) ? b : a; }   /* std::max(a,b) */
static double dmin(double a, double b) { return (b < a) ? b : a; }   /* std::min(a,b) */
static int imax(int a, int b) { return (a < b) ? b : a; }
static int imin(int a, int b) { return (b < a) ? b : a; }
static int iclamp(int v, int lo, int hi) { return (v < lo) ? lo : (hi < v) ? hi : v; } /* std::clamp */

/* std::min({a,b,c}) / std::max({a,b,c}) — our_gl.cpp:130-133 */
static double dmin3(double a, double b, double c) { double m = a; if (b < m) m = b; if (c < m) m = c; return m; }
static double dmax3(double a, double b, double c) { double m = a; if (m < b) m = b; if (m < c) m = c; return m; }

/* geometry.h:122-127 dot<n>: sum starts at 0, left to right */
static double dot3(const double a[3], const double b[3]) {
    double sum = 0;
    for (int i = 0; i < 3; ++i) sum += a[i] * b[i];
    return sum;
}
static double dot4(const double a[4], const double b[4]) {
    double sum = 0;
    for (int i = 0; i < 4; ++i) sum += a[i] * b[i];
    return sum;
}
/* geometry.h:136-140 normalized(): length==0 returns v unchanged, else v / length */
static void normalized3(const double v[3], double out[3]) {
    double length = sqrt(dot3(v, v));
    if (length == 0) { out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; return; }
    out[0] = v[0] / length; out[1] = v[1] / length; out[2] = v[2] / length;
}

void orc_init_viewport(double vp[16], int x, int y, int w, int h) {   /* our_gl.cpp:59-69 */
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) vp[r * 4 + c] = (r == c) ? 1.0 : 0.0;
    vp[0 * 4 + 0] = w / 2.0;
    vp[1 * 4 + 1] = h / 2.0;
    vp[0 * 4 + 3] = x + w / 2.0;
    vp[1 * 4 + 3] = y + h / 2.0;
    vp[2 * 4 + 2] = 1.0;
    vp[2 * 4 + 3] = 0.0;
}

void orc_stats_init(trgl_stats* s) {                                   /* our_gl.cpp:18-22 */
    s->triangles_rasterized = 0;
    s->fragments_drawn = 0;
    s->min_x = INT_MAX; s->min_y = INT_MAX; s->max_x = INT_MIN; s->max_y = INT_MIN;
    s->min_z = INFINITY; s->max_z = -INFINITY;
}

void orc_clear(orc_target* t, const uint8_t clear_bgra[4], double z_clear) {
    /* tgaimage.cpp:8-17: set(x,y,clear) for every pixel writes bgra[0..bpp) */
    static const uint8_t dflt[4] = { 0, 0, 0, 255 };                   /* tgaimage.h:33 */
    const uint8_t* c = clear_bgra ? clear_bgra : dflt;
    uint64_t npx = (uint64_t)t->w * (uint64_t)t->h;
    for (uint64_t i = 0; i < npx; ++i)
        for (int k = 0; k < t->bpp; ++k) t->fb[i * t->bpp + k] = c[k];
    for (uint64_t i = 0; i < npx; ++i) t->zbuf[i] = z_clear;            /* our_gl.cpp:72-74 */
}

/* ---- TGAColor (tgaimage.h:29-63) -------------------------------------------------------------- */
typedef struct { uint8_t bgra[4]; uint8_t bytespp; } color_t;

/* TGAColor(const uint8_t* p, uint8_t bpp) — tgaimage.h:46-50 */
static color_t color_from_ptr(const uint8_t* p, int bpp) {
    color_t c; c.bytespp = (uint8_t)bpp;
    for (int i = 0; i < bpp; ++i) c.bgra[i] = p[i];
    for (int i = bpp; i < 4; ++i) c.bgra[i] = 0;
    return c;
}
static color_t color_from_packed(uint32_t v) {        /* TGAColor(R,G,B,A) — tgaimage.h:35-38 */
    color_t c; c.bytespp = 4;
    c.bgra[0] = (uint8_t)(v & 0xff); c.bgra[1] = (uint8_t)((v >> 8) & 0xff);
    c.bgra[2] = (uint8_t)((v >> 16) & 0xff); c.bgra[3] = (uint8_t)((v >> 24) & 0xff);
    return c;
}
/* TGAColor::operator*(float) — tgaimage.h:55-62 */
static color_t color_scale(color_t c, float intensity) {
    color_t res = c;
    if (intensity < 0.f) intensity = 0.f;
    if (intensity > 1.f) intensity = 1.f;
    for (int i = 0; i < 4; i++) res.bgra[i] = (uint8_t)(c.bgra[i] * intensity);
    return res;
}

/* ---- samplers (model.cpp:415-459, TGAImage::get tgaimage.cpp:24-30) ---------------------------- */
static const orc_texture* tex_slot(const orc_texture* textures, int slot) {
    if (!textures || slot < 0 || slot >= TRGL_MAX_TEXTURES) return 0;
    if (!textures[slot].data || textures[slot].w <= 0) return 0;       /* hasX(): width() > 0 */
    return &textures[slot];
}
static color_t tex_fetch(const orc_texture* t, const double uv[2]) {
    int x = iclamp(x86_cvttsd2si(uv[0] * t->w), 0, t->w - 1);
    int y = iclamp(x86_cvttsd2si(uv[1] * t->h), 0, t->h - 1);
    return color_from_ptr(t->data + ((size_t)x + (size_t)y * t->w) * t->bpp, t->bpp);
}
/* the clamped nearest-texel fetch on its own, so that it can be pinned against the reference's compiled
 * IShader::sample2D (our_gl.h:38-44) + TGAImage::get (tgaimage.cpp:24-30): same int(uv * size) truncation, same clamp */
int orc_tex_fetch(const orc_texture* t, const double uv[2], uint8_t bgra[4]) {
    color_t c = tex_fetch(t, uv);
    for (int i = 0; i < 4; ++i) bgra[i] = c.bgra[i];
    return c.bytespp;
}
static color_t model_diffuse(const orc_texture* textures, int slot, const double uv[2]) {
    const orc_texture* t = tex_slot(textures, slot);                    /* model.cpp:415-426 */
    if (!t) { color_t c = { { 255, 255, 255, 255 }, 4 }; return c; }
    return tex_fetch(t, uv);
}
static void model_normal(const orc_texture* textures, int slot, const double uv[2], double out[3]) {
    const orc_texture* t = tex_slot(textures, slot);                    /* model.cpp:428-445 */
    if (!t) { out[0] = 0; out[1] = 0; out[2] = 1; return; }
    color_t c = tex_fetch(t, uv);
    double n[3];
    n[0] = (double)c.bgra[2] / 255.0 * 2.0 - 1.0;
    n[1] = (double)c.bgra[1] / 255.0 * 2.0 - 1.0;
    n[2] = (double)c.bgra[0] / 255.0 * 2.0 - 1.0;
    normalized3(n, out);
}
static float model_specular(const orc_texture* textures, int slot, const double uv[2]) {
    const orc_texture* t = tex_slot(textures, slot);                    /* model.cpp:447-459 */
    if (!t) return 1.0f;
    color_t c = tex_fetch(t, uv);
    return c.bgra[0] / 255.0f;
}

/* ---- varyings interpolation: v0*b0 + v1*b1 + v2*b2 per component (main.cpp:94-104) ------------- */
static void interp(const double* v0, const double* v1, const double* v2, const double b[3], int n, double* out) {
    for (int i = 0; i < n; ++i) out[i] = (v0[i] * b[0] + v1[i] * b[1]) + v2[i] * b[2];
}

/* (ModelView * vec4(n,0)).xyz() — main.cpp:116-119, geometry.h:186-192 */
static void mv_dir(const double mv[16], const double n[3], double out[3]) {
    double v4[4] = { n[0], n[1], n[2], 0.0 };
    for (int r = 0; r < 3; ++r) out[r] = dot4(mv + 4 * r, v4);
}

/* PhongShader::fragment — main.cpp:92-170 */
static color_t frag_phong(const trgl_uniforms* u, const orc_texture* tx, const double* vary, const double b[3]) {
    const double* uvv = vary;          /* varying_uv[3]           */
    const double* pos = vary + 6;      /* varying_position_eye[3] */
    const double* nrm = vary + 15;     /* varying_normal_eye[3]   */
    double position_eye[3], geometry_normal[3], uv[2];
    interp(pos, pos + 3, pos + 6, b, 3, position_eye);
    interp(nrm, nrm + 3, nrm + 6, b, 3, geometry_normal);
    interp(uvv, uvv + 2, uvv + 4, b, 2, uv);

    color_t base_color = model_diffuse(tx, u->tex_diffuse, uv);
    double specular_power = dmax(1.0, (double)model_specular(tx, u->tex_specular, uv));

    double brightness = (base_color.bgra[0] + base_color.bgra[1] + base_color.bgra[2]) / (3.0 * 255.0);
    int is_eye_pixel = (brightness >= 0.85) && (specular_power <= 5.0);

    double normal_map_value[3], normal_map_eye[3];
    model_normal(tx, u->tex_normal, uv, normal_map_value);
    mv_dir(u->model_view, normal_map_value, normal_map_eye);

    double final_normal[3];
    if (is_eye_pixel) {
        for (int i = 0; i < 3; ++i) final_normal[i] = geometry_normal[i];
    } else {
        double s = u->normal_map_strength, mix[3];
        for (int i = 0; i < 3; ++i) mix[i] = geometry_normal[i] * (1.0 - s) + normal_map_eye[i] * s;
        normalized3(mix, final_normal);
    }

    double neg_pos[3], view_direction[3];
    for (int i = 0; i < 3; ++i) neg_pos[i] = position_eye[i] * -1.0;   /* geometry.h:243-246 */
    normalized3(neg_pos, view_direction);

    const double* Lk = u->key_light_dir_eye;
    double key_diffuse = dmax(0.0, dot3(final_normal, Lk)) * 1.0;
    double key_specular = 0.0;
    {
        double k = 2.0 * dot3(final_normal, Lk), r[3], reflect_dir[3];
        for (int i = 0; i < 3; ++i) r[i] = final_normal[i] * k - Lk[i];
        normalized3(r, reflect_dir);
        double reflect_view_dot = dmax(0.0, dot3(reflect_dir, view_direction));
        key_specular = (reflect_view_dot > 0.0 ? pow(reflect_view_dot, specular_power) : 0.0) * 1.0;
    }
    double fill_diffuse = dmax(0.0, dot3(final_normal, u->fill_light_dir_eye)) * 0.35;
    double rim_diffuse = dmax(0.0, dot3(final_normal, u->rim_light_dir_eye)) * 0.6;

    double total_diffuse = key_diffuse + fill_diffuse + rim_diffuse;
    double total_specular = key_specular;
    double ambient = 0.10;

    color_t result = base_color;
    for (int ch = 0; ch < 3; ++ch) {
        double channel_value = base_color.bgra[ch];
        double final_value = channel_value * (ambient + total_diffuse) + 255.0 * (0.35 * total_specular);
        result.bgra[ch] = (unsigned char)dmin(255.0, final_value);
    }
    return result;
}

/* EyeShader::fragment — main.cpp:220-261 */
static color_t frag_eye(const trgl_uniforms* u, const orc_texture* tx, const double* vary, const double b[3]) {
    const double* uvv = vary;
    const double* pos = vary + 6;
    const double* nrm = vary + 15;
    double position_eye[3], n_interp[3], normal[3], uv[2];
    interp(pos, pos + 3, pos + 6, b, 3, position_eye);
    interp(nrm, nrm + 3, nrm + 6, b, 3, n_interp);
    normalized3(n_interp, normal);
    interp(uvv, uvv + 2, uvv + 4, b, 2, uv);

    color_t base_color = model_diffuse(tx, u->tex_diffuse, uv);
    double neg_pos[3], view_direction[3];
    for (int i = 0; i < 3; ++i) neg_pos[i] = position_eye[i] * -1.0;
    normalized3(neg_pos, view_direction);

    const double* Lk = u->key_light_dir_eye;
    double key_diffuse = dmax(0.0, dot3(normal, Lk)) * 1.0;
    double rim_diffuse = dmax(0.0, dot3(normal, u->rim_light_dir_eye)) * 0.6;
    double total_diffuse = key_diffuse + rim_diffuse;

    double specular_power = dmax(1.0, (double)model_specular(tx, u->tex_specular, uv)) * 8.0;
    double k = 2.0 * dot3(normal, Lk), r[3], reflect_dir[3];
    for (int i = 0; i < 3; ++i) r[i] = normal[i] * k - Lk[i];
    normalized3(r, reflect_dir);
    double reflect_view_dot = dmax(0.0, dot3(reflect_dir, view_direction));
    double specular = (reflect_view_dot > 0.0 ? pow(reflect_view_dot, specular_power) : 0.0);

    color_t result = base_color;
    for (int ch = 0; ch < 3; ++ch) {
        double channel_value = base_color.bgra[ch];
        double final_value = channel_value * (0.1 + total_diffuse) + 255.0 * (1.5 * specular);
        result.bgra[ch] = (unsigned char)dmin(255.0, final_value);
    }
    return result;
}

/* shader.fragment(bary) — the virtual up-call at our_gl.cpp:187, by kind (trgl.h); *discard = the first member of the pair it returns */
static color_t shade(int kind, const trgl_uniforms* un, const orc_texture* tx, const double* vary,
                     uint32_t packed_color, const double pc[3], int* discard) {
    *discard = 0;
    switch (kind) {
    case TRGL_SHADER_FLAT:
        return color_from_packed(packed_color);
    case TRGL_SHADER_CHECKER: {       /* the discarding test kind (trgl.h): cell parities of bar[0] and bar[1] differ -> discard */
        int a = x86_cvttsd2si(pc[0] * (double)un->reserved), c = x86_cvttsd2si(pc[1] * (double)un->reserved);
        *discard = ((a ^ c) & 1) != 0;
        return color_from_packed(packed_color);
    }
    case TRGL_SHADER_GOURAUD: {
        double intensity = (vary[0] * pc[0] + vary[1] * pc[1]) + vary[2] * pc[2];
        return color_scale(color_from_packed(packed_color), (float)intensity);
    }
    case TRGL_SHADER_PHONG: return frag_phong(un, tx, vary, pc);
    default:                return frag_eye(un, tx, vary, pc);
    }
}

/* barycentric() — our_gl.cpp:77-86 with cross() geometry.h:143-149 */
static void barycentric(const double A[2], const double B[2], const double C[2], const double P[2], double out[3]) {
    double s0[3] = { C[0] - A[0], B[0] - A[0], A[0] - P[0] };
    double s1[3] = { C[1] - A[1], B[1] - A[1], A[1] - P[1] };
    double u[3] = { s0[1] * s1[2] - s0[2] * s1[1],
                    s0[2] * s1[0] - s0[0] * s1[2],
                    s0[0] * s1[1] - s0[1] * s1[0] };
    if (fabs(u[2]) < 1e-12) { out[0] = -1; out[1] = 1; out[2] = 1; return; }
    out[0] = 1.0 - (u[0] + u[1]) / u[2];
    out[1] = u[1] / u[2];
    out[2] = u[0] / u[2];
}

/* rasterize() — our_gl.cpp:89-201 */
static void rasterize_one(orc_target* t, int kind, const trgl_uniforms* un, const orc_texture* tx,
                          const double* clip, const double* vary, uint32_t packed_color) {
    trgl_stats* st = &t->stats;
    ++st->triangles_rasterized;                                                       /* :90 */

    const double* v[3] = { clip, clip + 4, clip + 8 };
    if (v[0][3] <= 1e-12 || v[1][3] <= 1e-12 || v[2][3] <= 1e-12) return;            /* :94 */
    if (fabs(v[0][3]) < 1e-12 || fabs(v[1][3]) < 1e-12 || fabs(v[2][3]) < 1e-12) return; /* :97 (dead) */

    double ndc[3][4];
    for (int i = 0; i < 3; ++i) for (int c = 0; c < 4; ++c) ndc[i][c] = v[i][c] / v[i][3]; /* :101 */

    int z_out0 = (ndc[0][2] < -1.0 || ndc[0][2] > 1.0);                               /* :103-106 */
    int z_out1 = (ndc[1][2] < -1.0 || ndc[1][2] > 1.0);
    int z_out2 = (ndc[2][2] < -1.0 || ndc[2][2] > 1.0);
    if (z_out0 && z_out1 && z_out2) return;

    for (int i = 0; i < 3; ++i) for (int c = 0; c < 4; ++c) if (!isfinite(ndc[i][c])) return; /* :109-114 */

    double screen[3][2];                                                              /* :117-121 */
    for (int i = 0; i < 3; ++i) {
        screen[i][0] = dot4(t->viewport + 0, ndc[i]);
        screen[i][1] = dot4(t->viewport + 4, ndc[i]);
    }

    double e1x = screen[1][0] - screen[0][0], e1y = screen[1][1] - screen[0][1];      /* :124-127 */
    double e2x = screen[2][0] - screen[0][0], e2y = screen[2][1] - screen[0][1];
    double cross_product = e1x * e2y - e1y * e2x;
    if (cross_product <= 0) return;

    int min_x_px = imax(0, x86_cvttsd2si(floor(dmin3(screen[0][0], screen[1][0], screen[2][0]))));        /* :130-133 */
    int max_x_px = imin(t->w - 1, x86_cvttsd2si(ceil(dmax3(screen[0][0], screen[1][0], screen[2][0]))));
    int min_y_px = imax(0, x86_cvttsd2si(floor(dmin3(screen[0][1], screen[1][1], screen[2][1]))));
    int max_y_px = imin(t->h - 1, x86_cvttsd2si(ceil(dmax3(screen[0][1], screen[1][1], screen[2][1]))));
    if (min_x_px > max_x_px || min_y_px > max_y_px) return;                           /* :135 */

    st->min_x = imin(st->min_x, min_x_px);                                            /* :138-141 */
    st->min_y = imin(st->min_y, min_y_px);
    st->max_x = imax(st->max_x, max_x_px);
    st->max_y = imax(st->max_y, max_y_px);

    double w0 = v[0][3], w1 = v[1][3], w2 = v[2][3];                                  /* :144 */

    /* strip restriction (oracle extension for the multi-GPU tests): rows outside are skipped */
    int y_lo = imax(min_y_px, t->clip_y0), y_hi = imin(max_y_px, t->clip_y1 - 1);

    for (int x = min_x_px; x <= max_x_px; ++x) {                                      /* :147-148 */
        for (int y = y_lo; y <= y_hi; ++y) {
            double P[2] = { (double)x + 0.5, (double)y + 0.5 };                       /* :149 */
            double bc[3];
            barycentric(screen[0], screen[1], screen[2], P, bc);                      /* :150 */
            if (bc[0] < 0 || bc[1] < 0 || bc[2] < 0) continue;                        /* :152 */

            double z_ndc = bc[0] * ndc[0][2] + bc[1] * ndc[1][2] + bc[2] * ndc[2][2]; /* :156-158 */
            if (!isfinite(z_ndc)) continue;                                           /* :160 */

            size_t idx = (size_t)x + (size_t)y * (size_t)t->w;                        /* :162 */
            if (!(z_ndc < t->zbuf[idx])) continue;                                    /* :165 */

            double inv_w0 = (fabs(w0) > 1e-12) ? (1.0 / w0) : 0.0;                    /* :168-170 */
            double inv_w1 = (fabs(w1) > 1e-12) ? (1.0 / w1) : 0.0;
            double inv_w2 = (fabs(w2) > 1e-12) ? (1.0 / w2) : 0.0;
            double denom = bc[0] * inv_w0 + bc[1] * inv_w1 + bc[2] * inv_w2;          /* :172-174 */
            double pc[3];
            if (fabs(denom) < 1e-15) {                                                /* :177-185 */
                pc[0] = bc[0]; pc[1] = bc[1]; pc[2] = bc[2];
            } else {
                pc[0] = (bc[0] * inv_w0) / denom;
                pc[1] = (bc[1] * inv_w1) / denom;
                pc[2] = (bc[2] * inv_w2) / denom;
            }

            int discard;
            color_t color = shade(kind, un, tx, vary, packed_color, pc, &discard);    /* :187 */
            if (discard) continue;                                                    /* :188 */

            t->zbuf[idx] = z_ndc;                                                     /* :191 */
            for (int i = 0; i < t->bpp; ++i) t->fb[idx * t->bpp + i] = color.bgra[i]; /* :192, tgaimage.cpp:32-39 */

            ++st->fragments_drawn;                                                    /* :194 */
            st->min_z = dmin(st->min_z, z_ndc);                                       /* :197-198 */
            st->max_z = dmax(st->max_z, z_ndc);
        }
    }
}

int orc_fragment(int kind, const trgl_uniforms* u, const orc_texture* tx, const double* vary,
                 uint32_t packed_color, const double bary[3], uint8_t out_bgra[4]) {
    int discard;
    color_t c = shade(kind, u, tx, vary, packed_color, bary, &discard);
    memcpy(out_bgra, c.bgra, 4);
    return c.bytespp;
}
void orc_normalized3(const double v[3], double out[3]) { normalized3(v, out); }
void orc_mat4_mul_dir(const double m[16], const double n[3], double out[3]) { mv_dir(m, n, out); }
void orc_interp(const double* v0, const double* v1, const double* v2, const double b[3], int n, double* out) {
    interp(v0, v1, v2, b, n, out);
}

static int vary_count(int kind) {
    switch (kind) {
    case TRGL_SHADER_GOURAUD: return TRGL_VARY_GOURAUD;
    case TRGL_SHADER_PHONG:   return TRGL_VARY_PHONG;
    case TRGL_SHADER_EYE:     return TRGL_VARY_EYE;
    default:                  return 0;
    }
}

void orc_rasterize(orc_target* t, int kind, const trgl_uniforms* u, const orc_texture* textures,
                   const double* clip, const double* varyings, const uint32_t* colors, uint64_t n) {
    int K = vary_count(kind);
    for (uint64_t i = 0; i < n; ++i)
        rasterize_one(t, kind, u, textures, clip + 12 * i, K ? varyings + (size_t)K * i : 0,
                      colors ? colors[i] : 0xffffffffu);
}

uint64_t orc_fnv1a64(const void* p, uint64_t nbytes) {
    const uint8_t* b = (const uint8_t*)p;
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint64_t i = 0; i < nbytes; ++i) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}

/* TGAImage::write_tga_file + unload_rle_data — tgaimage.cpp:161-242 */
uint64_t orc_tga_encode(const uint8_t* data, int w, int h, int bpp, int vflip, int rle, uint8_t* out) {
    uint64_t n = 0;
    memset(out, 0, 18);                                   /* TGAHeader header = {} (tgaimage.h:10-25, packed) */
    out[2] = (uint8_t)(bpp == 1 ? (rle ? 11 : 3) : (rle ? 10 : 2));   /* datatypecode :175 */
    out[12] = (uint8_t)(w & 0xff); out[13] = (uint8_t)((w >> 8) & 0xff);
    out[14] = (uint8_t)(h & 0xff); out[15] = (uint8_t)((h >> 8) & 0xff);
    out[16] = (uint8_t)(bpp * 8);                          /* :169 */
    out[17] = vflip ? 0x00 : 0x20;                         /* :176 */
    n = 18;
    if (!rle) { memcpy(out + n, data, (size_t)w * h * bpp); return n + (uint64_t)w * h * bpp; }
    const int max_chunk_length = 128;                      /* :194 */
    int npixels = w * h, cur = 0;
    while (cur < npixels) {                                /* :198 */
        int chunkstart = cur * bpp, run_length = 1, raw = 1;
        while (cur + run_length < npixels && run_length < max_chunk_length) {     /* :203-213 */
            int equal = 1;
            for (int i = 0; i < bpp; i++) if (data[(cur + run_length) * bpp + i] != data[chunkstart + i]) { equal = 0; break; }
            if (!equal) break;
            run_length++; raw = 0;
        }
        if (!raw) {                                        /* :215-220 */
            out[n++] = (uint8_t)(run_length - 1 + 128);
            memcpy(out + n, data + chunkstart, bpp); n += bpp;
            cur += run_length;
        } else {                                           /* :221-238 */
            int rawstart = cur;
            run_length = 1;
            while (rawstart + run_length < npixels && run_length < max_chunk_length) {
                int next_equal = 1;
                for (int i = 0; i < bpp; i++)
                    if (data[(rawstart + run_length) * bpp + i] != data[(rawstart + run_length - 1) * bpp + i]) { next_equal = 0; break; }
                if (next_equal) break;
                run_length++;
            }
            out[n++] = (uint8_t)(run_length - 1);
            memcpy(out + n, data + rawstart * bpp, (size_t)run_length * bpp); n += (uint64_t)run_length * bpp;
            cur += run_length;
        }
    }
    return n;
}

/* ---- N1: vertex stage, main.cpp:71-90 (Phong) == main.cpp:199-218 (Eye) ------------------------------ */
void orc_vertex_stage(const double mv[16], const double proj[16], const double* vertices, int stride,
                      const uint32_t* indices, uint64_t nfaces, double* clip, double* varyings) {
    for (uint64_t f = 0; f < nfaces; ++f) {
        double* uv = varyings + 24 * f;        /* varying_uv[3]           */
        double* pe = uv + 6;                   /* varying_position_eye[3] */
        double* ne = uv + 15;                  /* varying_normal_eye[3]   */
        for (int v = 0; v < 3; ++v) {
            const double* vert = vertices + (size_t)indices[3 * f + v] * stride;     /* model.cpp:396-412 */
            double p4[4] = { vert[0], vert[1], vert[2], 1.0 }, n4[4] = { vert[3], vert[4], vert[5], 0.0 };
            double eye[4], nrm[4];
            for (int r = 0; r < 4; ++r) { eye[r] = dot4(mv + 4 * r, p4); nrm[r] = dot4(mv + 4 * r, n4); }   /* :77-86 */
            uv[2 * v] = vert[6]; uv[2 * v + 1] = vert[7];                              /* :75 */
            for (int k = 0; k < 3; ++k) { pe[3 * v + k] = eye[k]; ne[3 * v + k] = nrm[k]; }   /* :81,87 */
            for (int r = 0; r < 4; ++r) clip[12 * f + 4 * v + r] = dot4(proj + 4 * r, eye);   /* :89 */
        }
    }
}

/* ---- N4: main.cpp:269-311 ------------------------------------------------------------------------------ */
void orc_zbuffer_image(const double* zbuf, int w, int h, uint8_t* out) {
    double min_depth = 1e9, max_depth = -1e9;
    size_t n = (size_t)w * h;
    for (size_t i = 0; i < n; ++i) {
        double depth = zbuf[i];
        if (isfinite(depth)) { min_depth = dmin(min_depth, depth); max_depth = dmax(max_depth, depth); }
    }
    if (!isfinite(min_depth)) { memset(out, 255, n * 3); return; }           /* :284-292 (unreachable: 1e9 is finite) */
    if (max_depth - min_depth < 1e-7) max_depth = min_depth + 1e-7;           /* :294-296 */
    for (size_t i = 0; i < n; ++i) {
        double depth = zbuf[i];
        unsigned char value = 255;
        if (isfinite(depth)) {
            double normalized = (depth - min_depth) / (max_depth - min_depth);
            value = (unsigned char)(255.0 * (1.0 - normalized));
        }
        out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = value;
    }
}

/* main.cpp:317-362 */
static double compute_ssao_at(const double* zbuffer, int width, int height, int pixel_x, int pixel_y) {
    double center_depth = zbuffer[pixel_x + (size_t)pixel_y * width];
    if (!isfinite(center_depth)) return 1.0;
    int occluded_samples = 0, total_samples = 0;
    for (int direction = 0; direction < 8; ++direction) {
        double angle = 2.0 * M_PI * direction / 8;
        double dir_x = cos(angle), dir_y = sin(angle);
        for (int step = 1; step <= 8; ++step) {
            double radius = (double)step / 8 * 16.0;
            int sample_x = (int)round(pixel_x + dir_x * radius);
            int sample_y = (int)round(pixel_y + dir_y * radius);
            if (sample_x < 0 || sample_x >= width || sample_y < 0 || sample_y >= height) continue;
            double sample_depth = zbuffer[sample_x + (size_t)sample_y * width];
            if (!isfinite(sample_depth)) { total_samples++; continue; }
            if (sample_depth < center_depth - 1e-3) occluded_samples++;
            total_samples++;
        }
    }
    if (total_samples == 0) return 1.0;
    double occlusion_factor = (double)occluded_samples / (double)total_samples;
    return 1.0 - occlusion_factor * 0.35;
}
void orc_ssao(const double* zbuf, int w, int h, uint8_t* out) {               /* main.cpp:757-763 */
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        double ao_value = compute_ssao_at(zbuf, w, h, x, y);
        unsigned char intensity = (unsigned char)(255.0 * ao_value);
        size_t i = (size_t)x + (size_t)y * w;
        out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = intensity;
    }
}
void orc_composite(const uint8_t* fb, const uint8_t* ao, int w, int h, uint8_t* out) {   /* main.cpp:771-783 */
    size_t n = (size_t)w * h;
    for (size_t i = 0; i < n; ++i) {
        double ao_factor = ao[3 * i] / 255.0;
        for (int c = 0; c < 3; ++c) out[3 * i + c] = (unsigned char)dmin(255.0, (double)fb[3 * i + c] * ao_factor);
    }
}

/* TGAImage::read_tga_file + load_rle_data — tgaimage.cpp:76-160, on a file image in memory.  std::ifstream semantics are
 * spelled out: a short read delivers what is there and fails the stream; afterwards every read is a no-op.
 * Returns 1 where the reference returns true (w, h, bpp and w*h*bpp pixel bytes written), else 0. */
typedef struct { const uint8_t* p; size_t n, pos; int good; } orc_stream;
static void orc_read(orc_stream* s, uint8_t* dst, size_t k) {
    if (!s->good) return;
    size_t avail = s->pos < s->n ? s->n - s->pos : 0, m = k < avail ? k : avail;
    if (m) memcpy(dst, s->p + s->pos, m);
    s->pos += m;
    if (m < k) s->good = 0;
}
int orc_tga_decode(const uint8_t* file, uint64_t size, int* pw, int* ph, int* pbpp, uint8_t* data, uint64_t cap) {
    orc_stream in = { file, (size_t)size, 0, 1 };
    uint8_t hd[18] = { 0 };
    orc_read(&in, hd, 18);                                              /* :85 */
    if (!in.good) return 0;                                              /* :86-90 */
    int w = hd[12] | (hd[13] << 8), h = hd[14] | (hd[15] << 8), bpp = hd[16] >> 3;   /* :92-94 */
    *pw = w; *ph = h; *pbpp = bpp;
    if (w <= 0 || h <= 0 || (bpp != 1 && bpp != 3 && bpp != 4)) return 0;            /* :96-99 */
    if ((uint64_t)w * h * bpp > cap) return 0;
    memset(data, 0, (size_t)w * h * bpp);                                /* :101 data.resize() */
    in.pos += hd[0];                                                     /* :103 seekg(idlength, cur) */
    if (hd[2] == 2 || hd[2] == 3) {                                      /* :105-108 */
        orc_read(&in, data, (size_t)w * h * bpp);
    } else if (hd[2] == 10 || hd[2] == 11) {                             /* :109-112, load_rle_data :128-160 */
        int pixelcount = w * h, currentpixel = 0;
        uint8_t c[4] = { 0, 0, 0, 255 };                                 /* TGAColor c; tgaimage.h:33 */
        while (currentpixel < pixelcount) {
            uint8_t chunkheader = 0;
            orc_read(&in, &chunkheader, 1);
            if (chunkheader < 128) {
                int count = chunkheader + 1;
                for (int i = 0; i < count; i++) {
                    orc_read(&in, c, (size_t)bpp);
                    if (currentpixel >= pixelcount) return 0;            /* the reference writes out of bounds here, then fails (:145) */
                    for (int t = 0; t < bpp; t++) data[currentpixel * bpp + t] = c[t];
                    currentpixel++;
                }
            } else {
                int count = chunkheader - 127;
                orc_read(&in, c, (size_t)bpp);
                for (int i = 0; i < count; i++) {
                    if (currentpixel >= pixelcount) return 0;            /* :155 */
                    for (int t = 0; t < bpp; t++) data[currentpixel * bpp + t] = c[t];
                    currentpixel++;
                }
            }
        }
    } else return 0;                                                     /* :113-116 */
    int bytes_per_line = w * bpp;
    if (!(hd[17] & 0x20)) {                                              /* :118 flip_vertically, :59-71 */
        for (int y = 0; y < h / 2; y++)
            for (int k = 0; k < bytes_per_line; k++) {
                uint8_t t = data[y * bytes_per_line + k];
                data[y * bytes_per_line + k] = data[(h - 1 - y) * bytes_per_line + k];
                data[(h - 1 - y) * bytes_per_line + k] = t;
            }
    }
    if (hd[17] & 0x10) {                                                 /* :119 flip_horizontally, :42-57 */
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w / 2; x++)
                for (int i = 0; i < bpp; i++) {
                    uint8_t t = data[y * bytes_per_line + x * bpp + i];
                    data[y * bytes_per_line + x * bpp + i] = data[y * bytes_per_line + (w - 1 - x) * bpp + i];
                    data[y * bytes_per_line + (w - 1 - x) * bpp + i] = t;
                }
    }
    return 1;
}
