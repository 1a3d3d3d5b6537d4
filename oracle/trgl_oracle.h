/*
 * trgl_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement (plain C, fp64, no contraction) of the reference's rasterize() hot path:
 * /root/reference/our_gl.cpp:59-74,77-86,89-201 plus the fragment bodies main.cpp:92-170,220-261
 * and the samplers model.cpp:415-459.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product library never does.
 *
 * Pinning: rasterize()/barycentric()/TGAImage::set/TGAColor::operator* are pinned bit-for-bit
 * against the reference itself compiled from /root/reference (oracle/_ref, see oracle/Makefile
 * and tests/golden/).  The PHONG / EYE fragment bodies live in main.cpp, which cannot be compiled
 * here (model.h needs Assimp, absent): they are restated from the text and checked against the
 * same restatement run through the reference's own rasterize()+geometry.h — "parity unpinned"
 * against a compiled main.cpp.
 */
#ifndef TRGL_ORACLE_H
#define TRGL_ORACLE_H

#include <stdint.h>
#include "../include/trgl.h"   /* POD structs shared with the C ABI: trgl_uniforms, trgl_stats */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_texture {
    const uint8_t* data;   /* NULL = material has no such map */
    int w, h, bpp;
} orc_texture;

typedef struct orc_target {
    uint8_t* fb;           /* w*h*bpp bytes, TGAImage layout (tgaimage.cpp:32-39) */
    double*  zbuf;         /* w*h doubles (our_gl.cpp:72-74) */
    int      w, h, bpp;
    int      clip_y0, clip_y1;   /* rows rasterized: [clip_y0, clip_y1); whole image = [0,h) */
    double   viewport[16]; /* row-major mat<4,4> (our_gl.cpp:14) */
    trgl_stats stats;      /* our_gl.cpp:18-22 */
} orc_target;

/* our_gl.cpp:59-69 */
void orc_init_viewport(double vp[16], int x, int y, int w, int h);
/* our_gl.cpp:18-22 initial values */
void orc_stats_init(trgl_stats* s);
/* tgaimage.cpp:8-17 + our_gl.cpp:72-74 */
void orc_clear(orc_target* t, const uint8_t clear_bgra[4], double z_clear);

/* n calls of rasterize() (our_gl.cpp:89-201) in array order. */
void orc_rasterize(orc_target* t, int shader_kind, const trgl_uniforms* u,
                   const orc_texture* textures /* [TRGL_MAX_TEXTURES] or NULL */,
                   const double* clip, const double* varyings, const uint32_t* colors, uint64_t n);

/* One IShader::fragment() call (our_gl.cpp:187) of the given kind: bary is the perspective-correct
 * barycentric vector the rasterizer passes.  Returns bytespp of the colour (tgaimage.h:31). */
/* nearest-texel fetch with the samplers' index math (model.cpp:415-425 = our_gl.h:38-44); returns TGAColor::bytespp */
int orc_tex_fetch(const orc_texture* t, const double uv[2], uint8_t bgra[4]);
int orc_fragment(int shader_kind, const trgl_uniforms* u, const orc_texture* textures,
                 const double* varyings, uint32_t packed_color, const double bary[3], uint8_t out_bgra[4]);

/* value-math helpers exposed so tests can pin them against the reference's geometry.h */
void orc_normalized3(const double v[3], double out[3]);                       /* geometry.h:136-140 */
void orc_mat4_mul_dir(const double m[16], const double n[3], double out[3]);  /* geometry.h:186-192, w=0 */
void orc_interp(const double* v0, const double* v1, const double* v2, const double b[3], int n, double* out);

/* The bytes TGAImage::write_tga_file(name, vflip, rle) writes (tgaimage.cpp:161-242, header tgaimage.h:10-25).
 * out must hold 18 + w*h*bpp + w*h bytes (worst case); returns the length. */
uint64_t orc_tga_encode(const uint8_t* data, int w, int h, int bpp, int vflip, int rle, uint8_t* out);
/* TGAImage::read_tga_file + load_rle_data (tgaimage.cpp:76-160) on a file image in memory; 1 = the reference returns true. */
int orc_tga_decode(const uint8_t* file, uint64_t size, int* w, int* h, int* bpp, uint8_t* data, uint64_t cap);

/* ---- SURVEY.md §8(f) next rows, restated from main.cpp (unbuildable here: "parity unpinned" vs a compiled main.cpp) ---- */

/* N1: PhongShader::vertex / EyeShader::vertex (main.cpp:71-90,199-218) for every face-vertex of an indexed mesh.
 * vertices: nverts x stride doubles, position at +0, normal at +3, texcoord at +6 (model.h:14-20 Vertex);
 * out clip [nfaces][12], out varyings [nfaces][24] in the layout of include/trgl.h. */
void orc_vertex_stage(const double mv[16], const double proj[16], const double* vertices, int stride_doubles,
                      const uint32_t* indices, uint64_t nfaces, double* clip, double* varyings);

/* N4: save_zbuffer_image's pixels (main.cpp:269-311): out = w*h*3 bytes (B,G,R). */
void orc_zbuffer_image(const double* zbuf, int w, int h, uint8_t* out_bgr);
/* N4: the SSAO map (main.cpp:317-362,757-763) with the reference's constants; out = w*h*3 bytes. */
void orc_ssao(const double* zbuf, int w, int h, uint8_t* out_bgr);
/* N4: final composite (main.cpp:768-783): out = w*h*3 bytes from an RGB framebuffer and the AO map. */
void orc_composite(const uint8_t* fb_bgr, const uint8_t* ao_bgr, int w, int h, uint8_t* out_bgr);

/* FNV-1a 64 over raw bytes (used for fixtures) */
uint64_t orc_fnv1a64(const void* p, uint64_t nbytes);

#ifdef __cplusplus
}
#endif
#endif
