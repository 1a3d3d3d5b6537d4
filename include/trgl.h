/*
 * trgl.h — C ABI of the MI355X tile rasterizer (drop-in for the reference's rasterize() hot path).
 *
 * The reference (AnnaUshnova/tinyrenderder) has no FFI: its hot path is the C++ source-level
 * interface in our_gl.h.  Each entry point below names the reference interface it replaces
 * (file:line under the reference tree).  Plain pointers and sizes only; no C++ / torch types.
 *
 * Conventions
 *   - every function returns 0 on success, <0 (a TRGL_E_* code) on failure; the message is
 *     available from trgl_last_error().  Nothing ever throws across this boundary.
 *   - invalid triangles are silently dropped exactly as the reference does (our_gl.cpp:94-135);
 *     that is not an error.
 *   - one context per GPU, externally synchronised (the reference is single-threaded,
 *     our_gl.cpp:12-22 keeps all state in unsynchronised globals).
 *   - all arithmetic on the path is IEEE fp64 without contraction, in the reference's operation
 *     order; framebuffer bytes and z-buffer bits are identical to the reference's.
 */
#ifndef TRGL_H
#define TRGL_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRGL_VERSION 1
#define TRGL_MAX_TEXTURES 16

/* error codes */
#define TRGL_OK            0
#define TRGL_E_INVALID    -1   /* bad argument */
#define TRGL_E_HIP        -2   /* HIP runtime error (message in trgl_last_error) */
#define TRGL_E_NOMEM      -3
#define TRGL_E_STATE      -4   /* call not valid in the current state */
#define TRGL_E_UNSUPPORTED -5

/* where the pointers handed to trgl_draw() live */
#define TRGL_MEM_HOST   0      /* copied before trgl_draw returns */
#define TRGL_MEM_DEVICE 1      /* HBM-resident; must stay valid until trgl_flush has completed */

/*
 * Shader kinds: the device-side restatements of IShader::fragment() bodies (our_gl.h:51).
 * A C++ virtual cannot be called from a kernel, so a shader is a kind id + a POD uniform block +
 * per-triangle varyings snapshotted at rasterize() call time.
 *   FLAT    : one BGRA colour per triangle; fragment returns it unchanged.
 *   GOURAUD : three per-vertex intensities (K=3 doubles per triangle) and one BGRA base colour per
 *             triangle; fragment = base * (float)(i0*b0 + i1*b1 + i2*b2) with the semantics of
 *             TGAColor::operator*(float) (tgaimage.h:55-62).
 *   PHONG   : PhongShader::fragment (main.cpp:92-170), K=24 doubles per triangle.
 *   EYE     : EyeShader::fragment   (main.cpp:220-261), K=24 doubles per triangle.
 * Varyings layout for PHONG/EYE is the memory image of the shader's three member arrays
 * (main.cpp:47-49,181-183): uv[3] (6 doubles), position_eye[3] (9), normal_eye[3] (9).
 *
 *   CHECKER : the kind that DISCARDS (our_gl.h:51, our_gl.cpp:187-188: `if (discard) continue;` skips the depth write, the
 *             colour write and the counters).  One BGRA colour per triangle as for FLAT; fragment(bar) returns
 *             { (((int)(bar[0] * cells) ^ (int)(bar[1] * cells)) & 1) != 0, colour } with cells = uniforms->reserved and bar the
 *             perspective-correct barycentrics rasterize() passes (our_gl.cpp:168-185).  Its fragments are evaluated in
 *             submission order per pixel, like FLAT and GOURAUD ones.  (The reference ships no discarding shader; this is the
 *             plugin surface's discard path made testable: oracle/ref_harness.cpp holds the same IShader subclass.)
 *
 * PHONG / EYE are shaded once per visible pixel: with no discard and no side effects in fragment() (main.cpp:92-170,220-261
 * always return false), shading only the last fragment that passed the z-test at a pixel gives the framebuffer of shading
 * every z-pass in order.  A new kind whose fragment() can discard must be shaded in order, as CHECKER is.
 */
#define TRGL_SHADER_FLAT    0
#define TRGL_SHADER_GOURAUD 1
#define TRGL_SHADER_PHONG   2
#define TRGL_SHADER_EYE     3
#define TRGL_SHADER_CHECKER 4
#define TRGL_NUM_SHADERS    5

/* doubles of varyings per triangle for each kind */
#define TRGL_VARY_FLAT    0
#define TRGL_VARY_GOURAUD 3
#define TRGL_VARY_PHONG   24
#define TRGL_VARY_EYE     24
#define TRGL_VARY_CHECKER 0

/*
 * Uniform block for PHONG / EYE (ignored by FLAT / GOURAUD; may be NULL for those).
 * model_view is the value of the *global* ModelView when rasterize() was called: the reference
 * reads the global inside fragment() (main.cpp:116), so a deferred renderer must snapshot it.
 * Light directions are the shader members set by initLightDirections (main.cpp:55-69,187-197).
 * tex_* are texture slots filled by trgl_upload_texture, or -1 for "material has no such map"
 * (model.cpp:416-418,429-431,448-450 constants apply).
 */
typedef struct trgl_uniforms {
    double  model_view[16];           /* row-major mat<4,4> (geometry.h:154-156) */
    double  key_light_dir_eye[3];
    double  fill_light_dir_eye[3];    /* PHONG only */
    double  rim_light_dir_eye[3];
    double  normal_map_strength;      /* PHONG only (main.cpp:51) */
    int32_t tex_diffuse;
    int32_t tex_normal;               /* PHONG only */
    int32_t tex_specular;
    int32_t reserved;                 /* CHECKER: cells per barycentric axis (>= 1) */
} trgl_uniforms;

/* The reference's diagnostic counters (our_gl.cpp:18-22), per context instead of process-global. */
typedef struct trgl_stats {
    uint64_t triangles_rasterized;    /* every rasterize() call, our_gl.cpp:90 */
    uint64_t fragments_drawn;         /* every z-passing write, incl. later overwritten, :194 */
    int32_t  min_x, min_y, max_x, max_y;   /* union of clamped triangle bboxes, :138-141 */
    double   min_z, max_z;            /* range of written z, :197-198 */
} trgl_stats;

/* phases timed with HIP events on the context's stream when profiling is on */
#define TRGL_PHASE_SETUP   0   /* per-triangle setup + tile-overlap count */
#define TRGL_PHASE_BIN     1   /* scan + pair expansion + stable tile sort */
#define TRGL_PHASE_RASTER  2   /* tile raster (coverage, z-test, fragment) + tile flush */
#define TRGL_PHASE_TOTAL   3   /* first kernel to last kernel of a flush */
#define TRGL_PHASE_RASTER_KERNEL 4   /* the k_raster launch alone (RASTER also holds the work-item and counter-fold kernels) */
#define TRGL_NUM_PHASES    5

typedef struct trgl_ctx trgl_ctx;

/* ---- lifetime ------------------------------------------------------------------------------ */

/* Replaces: TGAImage framebuffer(W,H,bpp) (tgaimage.cpp:8-17) + init_zbuffer(W,H)
 * (our_gl.cpp:72-74) + init_viewport(0,0,W,H) (our_gl.cpp:59-69).  bpp is 1, 3 or 4
 * (TGAImage::Format, tgaimage.h:69).  The framebuffer starts cleared to TGAColor() = (0,0,0,255)
 * (tgaimage.h:33), the z-buffer to +inf, the stats to their initial values (our_gl.cpp:18-22). */
int trgl_create(int device, int width, int height, int bpp, trgl_ctx** out);
int trgl_destroy(trgl_ctx* ctx);
const char* trgl_last_error(const trgl_ctx* ctx /* may be NULL: creation errors */);

/* ---- pipeline state ------------------------------------------------------------------------ */

/* Replaces: the global `Viewport` (our_gl.cpp:14); m is the row-major 4x4. */
int trgl_set_viewport(trgl_ctx* ctx, const double m[16]);
/* Replaces: init_viewport(x,y,w,h) (our_gl.cpp:59-69). */
int trgl_init_viewport(trgl_ctx* ctx, int x, int y, int w, int h);
/* Replaces: TGAImage(w,h,bpp,clear) fill (tgaimage.cpp:8-17) and init_zbuffer (our_gl.cpp:72-74).
 * clear_bgra NULL = TGAColor(); z_clear is normally +infinity. */
int trgl_clear(trgl_ctx* ctx, const uint8_t clear_bgra[4], double z_clear);
/* Replaces: TGAImage textures held by Model::materials[0] (model.h:34-44); row-major, `bpp`
 * bytes per texel in B,G,R[,A] order, row 0 first, exactly TGAImage::buffer() (tgaimage.h:93). */
int trgl_upload_texture(trgl_ctx* ctx, int slot, const uint8_t* texels, int w, int h, int bpp);

/* Multi-GPU: restrict this context to framebuffer rows [y0,y1) (a horizontal strip).  Triangles
 * are still all counted/bboxed (setup is replicated), pixels outside the strip are not touched. */
int trgl_set_strip(trgl_ctx* ctx, int y0, int y1);

/* Multi-GPU, load-balanced alternative to one strip: the image is cut into bands of `band_rows` rows (a multiple of 32) that
 * are dealt round-robin to `world` contexts; this one takes the bands whose number is `rank` modulo `world`.  A mesh that sits
 * in the middle rows then loads every rank alike.  Within each period of world * band_rows rows the bands lie in rank order, so
 * one in-place all-gather per period joins them (tinyrenderder_amd/shard.py: gather_bands).  world = 1 or trgl_set_strip()
 * returns to a single strip. */
int trgl_set_interleave(trgl_ctx* ctx, int band_rows, int rank, int world);

/* Multi-GPU, the exchange step: join the rows that the `world` contexts of a render own (one process and one context per GPU,
 * trgl_set_strip with equal strips or trgl_set_interleave) into EVERY context's framebuffer - and z-buffer when `with_z` - by
 * in-place RCCL all-gathers over xGMI, queued on the context's stream behind the flush they follow (the call itself does not
 * wait: trgl_sync / trgl_read_framebuffer do).  One all-gather for strips, one per period of world * band_rows rows for bands.
 * This is the north-star's "RCCL all-gather of tile strips for the final TGAImage" behind the C ABI; a C++ host needs no RCCL
 * headers: `comm` comes from trgl_rccl_comm_create below (or is any ncclComm_t of `world` ranks the caller already has).
 * librccl.so.1 is loaded when first needed; TRGL_E_UNSUPPORTED if it is absent. */
int trgl_gather(trgl_ctx* ctx, void* nccl_comm, int rank, int world, int with_z);
/* The RCCL bootstrap for a C host: rank 0 obtains an id (128 bytes) and hands it to the other processes by whatever means
 * the launcher has (a file, a pipe, MPI); every rank then creates its communicator.  Wrap ncclGetUniqueId /
 * ncclCommInitRank / ncclCommDestroy. */
#define TRGL_RCCL_ID_BYTES 128
int trgl_rccl_unique_id(uint8_t id[TRGL_RCCL_ID_BYTES]);
int trgl_rccl_comm_create(const uint8_t id[TRGL_RCCL_ID_BYTES], int rank, int world, int device, void** nccl_comm);
int trgl_rccl_comm_destroy(void* nccl_comm);

/* ---- submission ---------------------------------------------------------------------------- */

/* Replaces: n consecutive calls of rasterize(clip, shader, framebuffer) (our_gl.h:58,
 * our_gl.cpp:89-201) with the same shader object.
 *   clip     : n x 12 doubles, the `Triangle` = vec<4>[3] memory image (our_gl.h:55)
 *   varyings : n x K doubles (K by kind, above) or NULL when K = 0
 *   colors   : n x uint32 (b | g<<8 | r<<16 | a<<24) for FLAT/GOURAUD, else NULL
 * Triangles are drawn in array order after everything submitted earlier (submission order is
 * observable: z ties keep the earlier triangle, our_gl.cpp:165).
 * n is not limited by the batching inside: a submission is cut into draws of 2^24 triangles and a new flush is
 * started before the 2^25-th triangle of one (neither shows in the frame or in the counters). */
int trgl_draw(trgl_ctx* ctx, int shader_kind, const trgl_uniforms* uniforms,
              const double* clip, const double* varyings, const uint32_t* colors,
              uint64_t n, int mem_kind);

/* SURVEY.md §8(f) N1 — the vertex stage on the device.
 * Replaces: the face loop `for v in 0..2: clip[v] = shader.vertex(face, v); rasterize(clip, shader, framebuffer)`
 * (main.cpp:660-666,692-698,715-721) with PhongShader::vertex / EyeShader::vertex (main.cpp:71-90,199-218):
 * eye = ModelView*(p,1), normal_eye = ModelView*(n,0), clip = projection*eye, for an indexed mesh.
 *   uniforms->model_view : the global ModelView (used by vertex AND fragment stage, as in the reference)
 *   projection           : the global Perspective, row-major 4x4
 *   vertices             : n_vertices x vertex_stride doubles; position at +0, normal at +3, texcoord at +6
 *                          (the reference's `Vertex`, model.h:14-20, has stride 14)
 *   indices              : 3*n_faces uint32 (Model::indices, model.h:115)
 * shader_kind is TRGL_SHADER_PHONG or TRGL_SHADER_EYE.  Host arrays are copied before return. */
int trgl_draw_indexed(trgl_ctx* ctx, int shader_kind, const trgl_uniforms* uniforms, const double projection[16],
                      const double* vertices, int vertex_stride, uint64_t n_vertices,
                      const uint32_t* indices, uint64_t n_faces, int mem_kind);

/* Execute everything submitted so far (asynchronously on the context's stream). */
int trgl_flush(trgl_ctx* ctx);
/* The same in two halves, for a caller that overlaps something with the first one: trgl_flush_begin runs per-triangle
 * setup and tile binning (neither reads nor writes the framebuffer / z-buffer), trgl_flush_end the tile raster.  Every
 * other entry point completes a begun flush first.  bench.py: the RCCL gather of the previous frame's strips runs
 * under the next frame's first half. */
int trgl_flush_begin(trgl_ctx* ctx);
int trgl_flush_end(trgl_ctx* ctx);
/* Wait for the context's stream. */
int trgl_sync(trgl_ctx* ctx);

/* ---- results ------------------------------------------------------------------------------- */

/* Replaces: TGAImage::buffer() / get() (tgaimage.h:93, tgaimage.cpp:24-30): W*H*bpp bytes,
 * index (x + y*W)*bpp, B,G,R[,A].  Implies flush + sync. */
int trgl_read_framebuffer(trgl_ctx* ctx, uint8_t* dst);
int trgl_write_framebuffer(trgl_ctx* ctx, const uint8_t* src);
/* Replaces: direct access to the global std::vector<double> zbuffer (our_gl.h:20;
 * main.cpp:700,730,751,759): W*H doubles, index x + y*W. */
int trgl_read_zbuffer(trgl_ctx* ctx, double* dst);
int trgl_write_zbuffer(trgl_ctx* ctx, const double* src);
/* Replaces: print_render_stats() (our_gl.cpp:204-210), as a struct. Implies flush + sync. */
int trgl_get_stats(trgl_ctx* ctx, trgl_stats* out);
int trgl_reset_stats(trgl_ctx* ctx);
/* Formats exactly the line print_render_stats() writes to stderr (our_gl.cpp:205-209). */
int trgl_format_stats(const trgl_stats* s, char* buf, size_t buflen);

/* Device-resident results, for collectives (RCCL all-gather of strips) and on-device consumers.
 * Valid until trgl_destroy. Rows outside the context's strip hold stale/cleared data. */
void* trgl_framebuffer_device_ptr(trgl_ctx* ctx);
void* trgl_zbuffer_device_ptr(trgl_ctx* ctx);
/* The hipStream_t all work of this context is enqueued on. */
void* trgl_stream(trgl_ctx* ctx);
/* Enqueue on the caller's hipStream_t instead (e.g. the stream a collective library orders against).  A NULL
 * handle is the legacy default stream — a real stream, and what torch's current stream usually is — so returning
 * to the context's own stream is use_own != 0.  The caller keeps its stream alive.  Implies a sync. */
int trgl_set_stream(trgl_ctx* ctx, void* hip_stream, int use_own);

/* ---- measurement --------------------------------------------------------------------------- */

int trgl_set_profiling(trgl_ctx* ctx, int on);
/* Cumulative milliseconds per phase since the last reset and the number of flushes measured. */
int trgl_get_phase_ms(trgl_ctx* ctx, double ms[TRGL_NUM_PHASES], uint64_t* flushes);
int trgl_reset_phase_ms(trgl_ctx* ctx);
/* Implementation traffic counters of the last flush: tri-tile pairs produced by binning. */
int trgl_get_last_flush_info(trgl_ctx* ctx, uint64_t* triangles, uint64_t* pairs, uint64_t* tiles);

/* ---- post-process on the resident z-buffer ----------------------------------- */

/* The reference's SSAO constants (main.cpp:317-321); trgl_ssao_defaults fills them in. */
typedef struct trgl_ssao_params {
    int32_t num_directions;        /* AO_NUM_DIRECTIONS = 8 (at most 16) */
    int32_t steps_per_direction;   /* AO_STEPS_PER_DIRECTION = 8 */
    double  sample_radius;         /* AO_SAMPLE_RADIUS = 16.0 px */
    double  occlusion_threshold;   /* AO_OCCLUSION_THRESHOLD = 1e-3 */
    double  intensity;             /* AO_INTENSITY = 0.35 */
} trgl_ssao_params;
void trgl_ssao_defaults(trgl_ssao_params* p);

/* Replaces: save_zbuffer_image's pixel loop (main.cpp:269-311), the SSAO loop (main.cpp:317-362,757-763) and the
 * final composite (main.cpp:768-783), computed on the device from the context's z-buffer and framebuffer (no
 * z-buffer readback).  Each output is W*H*3 bytes (B,G,R) in host memory and may be NULL; `final` needs `ao` to be
 * computed too (it is, internally).  params NULL = the reference's constants.  Implies flush + sync. */
int trgl_postprocess(trgl_ctx* ctx, const trgl_ssao_params* params, uint8_t* zbuffer_image, uint8_t* ao_map, uint8_t* final_image);

/* ---- TGA writer and reader (host only) ------------------------------------- */

/* Replaces: TGAImage::write_tga_file(name, vflip, rle) (tgaimage.cpp:161-242): produces exactly the bytes the
 * reference writes — 18-byte header (tgaimage.h:10-25), no footer, its RLE packetisation.  `out` needs
 * trgl_tga_max_size(w,h,bpp) bytes; *out_len receives the file length.  Needs no GPU and no context. */
size_t trgl_tga_max_size(int w, int h, int bpp);
int trgl_tga_encode(const uint8_t* pixels, int w, int h, int bpp, int vflip, int rle, uint8_t* out, size_t* out_len);

/* Replaces: TGAImage::read_tga_file + load_rle_data (tgaimage.cpp:76-160) on a .tga file image held in memory - the maps
 * sampled at model.cpp:415-459 reach the path through it.  trgl_tga_info parses the 18-byte header; trgl_tga_decode
 * fills width*height*bpp bytes in TGAImage::buffer() order (after the origin flips of tgaimage.cpp:118-119), with the
 * reference's treatment of truncated files (missing raw bytes stay 0, a cut RLE stream repeats the last colour).
 * Both return TRGL_E_INVALID where the reference returns false.  Needs no GPU and no context. */
int trgl_tga_info(const uint8_t* file, size_t size, int* width, int* height, int* bpp);
int trgl_tga_decode(const uint8_t* file, size_t size, uint8_t* pixels);

/* ---- OBJ reader (host only) -------------------------------------------------- */

/* Stands in for the Assimp import of model.cpp:89-205 (fan triangulation, FlipUVs, float precision, one vertex per
 * distinct v/vt/vn triple, the normal fallback of model.cpp:269-316).  On success *vertices holds *n_vertices records
 * of 14 doubles (the reference's `Vertex`, model.h:14-20) and *indices 3 * *n_faces uint32, both owned by the library
 * until trgl_obj_free.  Assimp's own vertex/face reordering is not reproducible: parity unpinned (tinyrenderder_amd/
 * shim/trgl_obj.h).  Needs no GPU. */
int trgl_obj_load(const char* path, double** vertices, uint64_t* n_vertices, uint32_t** indices, uint64_t* n_faces);
void trgl_obj_free(double* vertices, uint32_t* indices);

/* Self-test of the kernel's two exactness shortcuts (division by the per-triangle constant u.z through a
 * correctly rounded reciprocal + FMA corrections, and the division-free coverage signs) against the GPU's own
 * IEEE fp64 division, on `samples` random and adversarial operand pairs (all-ones significands, quotients next to
 * rounding midpoints, numerators next to u.z, signed zeros).  *mismatches must come back 0. */
int trgl_selftest_division(trgl_ctx* ctx, uint64_t samples, uint64_t seed, uint64_t* mismatches);

/*
 * Diagnostics: the samplers' nearest-texel fetch (Model::diffuse / normal / specular, model.cpp:415-459:
 * clamp(int(uv * size), 0, size - 1) then TGAImage::get, tgaimage.cpp:24-30) of texture `slot` at n host-side uv pairs;
 * out receives 5 bytes per sample: bgra[4], TGAColor::bytespp.  An empty slot samples as opaque white (model.cpp:416-418).
 * The tests compare it with what the reference's compiled IShader::sample2D (our_gl.h:38-44) returns.
 */
int trgl_selftest_sampler(trgl_ctx* ctx, int slot, const double* uv, uint64_t n, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif /* TRGL_H */
