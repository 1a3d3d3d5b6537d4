// trgl_api.cpp — host side of the C ABI in include/trgl.h: context, HBM buffers, the flush pipeline.
//
// A flush runs, on the context's own HIP stream:
//   setup (per draw) -> scan(counts) -> expand -> stable radix passes by tile id -> bounds -> raster
// and is the batched equivalent of the reference's per-face loop of rasterize() calls
// (main.cpp:660-666): submission order is preserved per tile, so results are identical.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "../../include/trgl.h"
#include "launch.h"
#include "trgl_device.h"
#include "../shim/trgl_image.h"
#include "../shim/trgl_obj.h"

using namespace trgl;

static thread_local std::string g_create_error;

struct StageChunk { char* base; size_t cap, used; };

struct trgl_ctx {
    int device = 0;
    int num_cus = 256;          // multiProcessorCount of the device
    int W = 0, H = 0, bpp = 0, tiles_x = 0, tiles_y = 0;
    hipStream_t stream = nullptr;       // the stream in use
    hipStream_t own_stream = nullptr;   // created with the context
    uint8_t* fb = nullptr;
    double* zb = nullptr;
    double vp[16];
    bool clear_pending = true;
    uint32_t clear_color = 0xff000000u;
    double clear_z = std::numeric_limits<double>::infinity();
    int strip_y0 = 0, strip_y1 = 0;
    int il_tiles = 0, il_world = 1, il_rank = 0;     // interleaved bands instead of one strip (trgl_set_interleave)

    DevTexture tex_host[TRGL_MAX_TEXTURES];
    DevTexture* tex_dev = nullptr;

    std::vector<DrawDesc> draws;
    uint64_t queued_tris = 0;
    std::vector<StageChunk> stage;
    int stage_hold = 0;                 // >0 while a draw call has staged data that no DrawDesc references yet

    TriRec* recs = nullptr; TriW* recs_w = nullptr; uint32_t* cnt = nullptr; uint2* tilebox = nullptr;
    // a flush whose first half (setup + binning) has run and whose raster half is still to be launched (trgl_flush_begin)
    struct { bool active = false; FrameParams fp; int flush_kind = 0; uint32_t cap = 0; int cur = 0; uint64_t N = 0; bool binned = false; } rp;
    hipEvent_t ev_pairs = nullptr;      // recorded behind the copy of the flush's pair count into pinned memory
    uint32_t* idbuf = nullptr; size_t cap_idbuf = 0;        // visibility buffer of PHONG / EYE flushes, [H][W]
    uint8_t* pp_out = nullptr; size_t cap_pp = 0;           // trgl_postprocess: three [H][W][3] images + two 64-bit z-range keys, kept between calls
    uint32_t* blk_sums = nullptr; size_t cap_blk = 0;       // pairs per setup block of 256 triangles
    uint32_t* chunk_off = nullptr; size_t cap_chunk = 0;    // pairs before every 16th setup block
    size_t cap_tris = 0;
    uint32_t* keys[2] = { nullptr, nullptr }; uint32_t* vals[2] = { nullptr, nullptr }; uint16_t* bmask[2] = { nullptr, nullptr };
    size_t cap_pairs = 0;
    uint32_t* hist = nullptr; size_t cap_hist = 0;
    uint32_t* scan_tmp = nullptr; size_t cap_scan = 0;
    uint32_t* tile_start = nullptr; uint32_t* tile_end = nullptr;
    uint4* items = nullptr; size_t cap_items = 0; uint32_t* n_items = nullptr;
    unsigned long long* item_stats = nullptr; size_t cap_item_stats = 0;
    DrawDesc* draws_dev = nullptr;
    DrawDesc* draws_pinned = nullptr;
    DevStats* stats_dev = nullptr;
    DevStats* stats_pinned = nullptr;

    uint64_t triangles_total = 0;       // our_gl.cpp:90 counts every call, host side
    uint64_t last_tris = 0, last_pairs = 0;

    bool profiling = false, events_pending = false;
    hipEvent_t ev[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    double phase_ms[TRGL_NUM_PHASES] = { 0, 0, 0, 0, 0 };
    uint64_t flushes_timed = 0;

    std::string err;
};

#define HIPCHK(ctx, expr)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                    \
            return TRGL_E_HIP;                                                                 \
        }                                                                                      \
    } while (0)

#define CHKCTX(ctx) do { if (!(ctx)) return TRGL_E_INVALID; if (hipSetDevice((ctx)->device) != hipSuccess) return TRGL_E_HIP; } while (0)

static int fail(trgl_ctx* c, int code, const char* msg) { c->err = msg; return code; }

static unsigned long long zkey_host(double d) {
    unsigned long long b; std::memcpy(&b, &d, 8);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
static double zkey_decode(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double d; std::memcpy(&d, &b, 8); return d;
}

static int reset_dev_stats(trgl_ctx* c) {
    DevStats s;
    s.fragments = 0;
    s.zmin_key = zkey_host(std::numeric_limits<double>::infinity());
    s.zmax_key = zkey_host(-std::numeric_limits<double>::infinity());
    s.min_x = INT32_MAX; s.min_y = INT32_MAX; s.max_x = INT32_MIN; s.max_y = INT32_MIN;
    s.pairs_total = 0; s.literal_tris = 0; s.large_tris = 0;
    s.zero_pos_key = s.zero_neg_key = TRGL_ZERO_KEY_EMPTY;
    s.zero_locked = 0; s.zero_sign = 0;
    for (int k = 0; k < 16; ++k) s.dbg[k] = 0;
    *c->stats_pinned = s;
    HIPCHK(c, hipMemcpyAsync(c->stats_dev, c->stats_pinned, sizeof(DevStats), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return TRGL_OK;
}

// (re)allocate a device buffer; the old contents are not kept
static int realloc_dev(trgl_ctx* c, void** p, size_t bytes) {
    if (*p) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(*p)); *p = nullptr; }
    HIPCHK(c, hipMalloc(p, bytes));
    return TRGL_OK;
}
template <class T> static int grow(trgl_ctx* c, T*& p, size_t& cap, size_t need) {
    if (need <= cap && p) return TRGL_OK;
    size_t ncap = need + need / 4 + 1024;
    int r = realloc_dev(c, (void**)&p, ncap * sizeof(T)); if (r) return r;
    cap = ncap;
    return TRGL_OK;
}

extern "C" {

const char* trgl_last_error(const trgl_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int trgl_create(int device, int width, int height, int bpp, trgl_ctx** out) {
    if (!out || width <= 0 || height <= 0 || width > 65535 || height > 65535 || !(bpp == 1 || bpp == 3 || bpp == 4)) {
        g_create_error = "trgl_create: bad arguments (width/height in 1..65535, bpp in {1,3,4})";
        return TRGL_E_INVALID;
    }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return TRGL_E_HIP; }
    trgl_ctx* c = new trgl_ctx();
    c->device = device; c->W = width; c->H = height; c->bpp = bpp;
    c->tiles_x = (width + TRGL_TILE - 1) / TRGL_TILE;
    c->tiles_y = (height + TRGL_TILE - 1) / TRGL_TILE;
    c->strip_y0 = 0; c->strip_y1 = height;
    { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n > 0) c->num_cus = n; }
    std::memset(c->tex_host, 0, sizeof(c->tex_host));
    size_t npx = (size_t)width * height, ntiles = (size_t)c->tiles_x * c->tiles_y;
#define CRE(expr) do { hipError_t e2 = (expr); if (e2 != hipSuccess) { g_create_error = std::string(#expr) + ": " + hipGetErrorString(e2); trgl_destroy(c); return TRGL_E_HIP; } } while (0)
    CRE(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    CRE(hipMalloc((void**)&c->fb, npx * bpp));
    CRE(hipMalloc((void**)&c->zb, npx * sizeof(double)));
    CRE(hipMalloc((void**)&c->tex_dev, sizeof(c->tex_host)));
    CRE(hipMemset(c->tex_dev, 0, sizeof(c->tex_host)));
    CRE(hipMalloc((void**)&c->tile_start, ntiles * 8 + 16));   // tile_start[ntiles] followed by tile_end[ntiles]: cleared together per flush (in 16-byte words)
    c->tile_end = c->tile_start + ntiles;
    CRE(hipMalloc((void**)&c->n_items, 8));                     // work items of the flush
    CRE(hipMemset(c->n_items, 0, 8));                           // k_fold_stats leaves it at 0 for the next flush
    CRE(hipMalloc((void**)&c->draws_dev, sizeof(DrawDesc) * TRGL_MAX_DRAWS));
    CRE(hipHostMalloc((void**)&c->draws_pinned, sizeof(DrawDesc) * TRGL_MAX_DRAWS));
    CRE(hipMalloc((void**)&c->stats_dev, sizeof(DevStats)));
    CRE(hipHostMalloc((void**)&c->stats_pinned, sizeof(DevStats)));
    for (int i = 0; i < 6; ++i) CRE(hipEventCreate(&c->ev[i]));
    CRE(hipEventCreateWithFlags(&c->ev_pairs, hipEventDisableTiming));
#undef CRE
    // init_viewport(0,0,W,H), our_gl.cpp:59-69
    trgl_init_viewport(c, 0, 0, width, height);
    if (reset_dev_stats(c) != TRGL_OK) { g_create_error = c->err; trgl_destroy(c); return TRGL_E_HIP; }
    *out = c;
    return TRGL_OK;
}

int trgl_destroy(trgl_ctx* c) {
    if (!c) return TRGL_E_INVALID;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& s : c->stage) (void)hipFree(s.base);
    for (int i = 0; i < TRGL_MAX_TEXTURES; ++i) if (c->tex_host[i].data) (void)hipFree((void*)c->tex_host[i].data);
    void* ptrs[] = { c->fb, c->zb, c->tex_dev, c->recs, c->recs_w, c->bmask[0], c->bmask[1], c->cnt, c->idbuf, c->pp_out, c->blk_sums, c->chunk_off, c->tilebox, c->keys[0], c->keys[1], c->vals[0],
                     c->vals[1], c->hist, c->scan_tmp, c->tile_start, c->draws_dev, c->stats_dev, c->items, c->n_items, c->item_stats };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (c->draws_pinned) (void)hipHostFree(c->draws_pinned);
    if (c->stats_pinned) (void)hipHostFree(c->stats_pinned);
    for (int i = 0; i < 6; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->ev_pairs) (void)hipEventDestroy(c->ev_pairs);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return TRGL_OK;
}

int trgl_set_viewport(trgl_ctx* c, const double m[16]) {
    CHKCTX(c);
    if (c->rp.active) { int fr = trgl_flush_end(c); if (fr) return fr; }
    if (!m) return fail(c, TRGL_E_INVALID, "trgl_set_viewport: null matrix");
    if (!c->draws.empty() && std::memcmp(c->vp, m, sizeof(c->vp)) != 0) {   // rasterize() reads Viewport at call time
        int r = trgl_flush(c); if (r) return r;
    }
    std::memcpy(c->vp, m, sizeof(c->vp));
    return TRGL_OK;
}

int trgl_init_viewport(trgl_ctx* c, int x, int y, int w, int h) {          // our_gl.cpp:59-69
    if (!c) return TRGL_E_INVALID;
    double m[16];
    for (int r = 0; r < 4; ++r) for (int k = 0; k < 4; ++k) m[4 * r + k] = (r == k) ? 1.0 : 0.0;
    m[0] = w / 2.0; m[5] = h / 2.0; m[3] = x + w / 2.0; m[7] = y + h / 2.0; m[10] = 1.0; m[11] = 0.0;
    if (c->stream == nullptr || c->draws.empty()) { std::memcpy(c->vp, m, sizeof(m)); return TRGL_OK; }
    return trgl_set_viewport(c, m);
}

int trgl_clear(trgl_ctx* c, const uint8_t bgra[4], double z_clear) {
    CHKCTX(c);
    if (c->rp.active) { int fr = trgl_flush_end(c); if (fr) return fr; }
    if (!c->draws.empty()) { int r = trgl_flush(c); if (r) return r; }      // earlier draws come first
    static const uint8_t dflt[4] = { 0, 0, 0, 255 };                         // TGAColor(), tgaimage.h:33
    const uint8_t* p = bgra ? bgra : dflt;
    c->clear_color = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    c->clear_z = z_clear;
    c->clear_pending = true;
    return TRGL_OK;
}

int trgl_upload_texture(trgl_ctx* c, int slot, const uint8_t* texels, int w, int h, int bpp) {
    CHKCTX(c);
    if (c->rp.active) { int fr = trgl_flush_end(c); if (fr) return fr; }
    if (slot < 0 || slot >= TRGL_MAX_TEXTURES) return fail(c, TRGL_E_INVALID, "trgl_upload_texture: bad slot");
    if (!c->draws.empty()) { int r = trgl_flush(c); if (r) return r; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->tex_host[slot].data) { HIPCHK(c, hipFree((void*)c->tex_host[slot].data)); c->tex_host[slot] = DevTexture{ nullptr, 0, 0, 0, 0 }; }
    if (texels && w > 0 && h > 0) {
        if (!(bpp == 1 || bpp == 3 || bpp == 4)) return fail(c, TRGL_E_INVALID, "trgl_upload_texture: bpp must be 1, 3 or 4");
        uint8_t* d = nullptr; size_t bytes = (size_t)w * h * bpp;
        HIPCHK(c, hipMalloc((void**)&d, bytes + 8));         // +8: the samplers read each texel as one 4-byte load
        HIPCHK(c, hipMemset(d + bytes, 0, 8));
        HIPCHK(c, hipMemcpy(d, texels, bytes, hipMemcpyHostToDevice));
        c->tex_host[slot] = DevTexture{ d, w, h, bpp, 0 };
    }
    HIPCHK(c, hipMemcpy(c->tex_dev, c->tex_host, sizeof(c->tex_host), hipMemcpyHostToDevice));
    return TRGL_OK;
}

int trgl_set_strip(trgl_ctx* c, int y0, int y1) {
    CHKCTX(c);
    if (c->rp.active) { int fr = trgl_flush_end(c); if (fr) return fr; }
    if (y0 < 0 || y1 > c->H || y0 > y1) return fail(c, TRGL_E_INVALID, "trgl_set_strip: need 0 <= y0 <= y1 <= H");
    if (!c->draws.empty()) { int r = trgl_flush(c); if (r) return r; }
    c->strip_y0 = y0; c->strip_y1 = y1;
    c->il_tiles = 0; c->il_world = 1; c->il_rank = 0;
    return TRGL_OK;
}

int trgl_set_interleave(trgl_ctx* c, int band_rows, int rank, int world) {
    CHKCTX(c);
    if (c->rp.active) { int fr = trgl_flush_end(c); if (fr) return fr; }
    if (band_rows <= 0 || band_rows % TRGL_TILE || world < 1 || rank < 0 || rank >= world)
        return fail(c, TRGL_E_INVALID, "trgl_set_interleave: band_rows must be a positive multiple of 32, 0 <= rank < world");
    if (!c->draws.empty()) { int r = trgl_flush(c); if (r) return r; }
    c->strip_y0 = 0; c->strip_y1 = c->H;
    c->il_tiles = world > 1 ? band_rows / TRGL_TILE : 0; c->il_world = world; c->il_rank = rank;
    return TRGL_OK;
}

static int vary_count(int kind) {
    switch (kind) {
    case TRGL_SHADER_GOURAUD: return TRGL_VARY_GOURAUD;
    case TRGL_SHADER_PHONG: return TRGL_VARY_PHONG;
    case TRGL_SHADER_EYE: return TRGL_VARY_EYE;
    case TRGL_SHADER_CHECKER: return TRGL_VARY_CHECKER;
    default: return 0;
    }
}

// device memory from the per-flush arena; chunks live until the flush that uses them is done
static int stage_alloc(trgl_ctx* c, size_t bytes, void** dev) {
    size_t need = (bytes + 255) & ~size_t(255);
    StageChunk* ch = nullptr;
    for (auto& s : c->stage) if (s.cap - s.used >= need) { ch = &s; break; }
    if (!ch) {
        StageChunk s; s.cap = need > (size_t(64) << 20) ? need : (size_t(64) << 20); s.used = 0; s.base = nullptr;
        HIPCHK(c, hipMalloc((void**)&s.base, s.cap));
        c->stage.push_back(s); ch = &c->stage.back();
    }
    *dev = ch->base + ch->used; ch->used += need;
    return TRGL_OK;
}
// copy host data into the arena
static int stage_copy(trgl_ctx* c, const void* src, size_t bytes, void** dev) {
    int r = stage_alloc(c, bytes, dev); if (r) return r;
    HIPCHK(c, hipMemcpy(*dev, src, bytes, hipMemcpyHostToDevice));   // "copied before trgl_draw returns"
    return TRGL_OK;
}

int trgl_draw(trgl_ctx* c, int kind, const trgl_uniforms* u, const double* clip, const double* vary,
              const uint32_t* colors, uint64_t n, int mem_kind) {
    CHKCTX(c);
    if (c->rp.active) { int fr = trgl_flush_end(c); if (fr) return fr; }
    if (kind < 0 || kind >= TRGL_NUM_SHADERS) return fail(c, TRGL_E_INVALID, "trgl_draw: unknown shader kind");
    if (n == 0) return TRGL_OK;
    if (!clip) return fail(c, TRGL_E_INVALID, "trgl_draw: clip is null");
    int K = vary_count(kind);
    if (K && !vary) return fail(c, TRGL_E_INVALID, "trgl_draw: this shader kind needs varyings");
    if ((kind == TRGL_SHADER_PHONG || kind == TRGL_SHADER_EYE) && !u) return fail(c, TRGL_E_INVALID, "trgl_draw: PHONG/EYE need uniforms");
    if (kind == TRGL_SHADER_CHECKER && (!u || u->reserved < 1)) return fail(c, TRGL_E_INVALID, "trgl_draw: CHECKER needs uniforms with reserved = cells >= 1");
    if (mem_kind != TRGL_MEM_HOST && mem_kind != TRGL_MEM_DEVICE) return fail(c, TRGL_E_INVALID, "trgl_draw: bad mem_kind");
    if (c->queued_tris + n > 0xffffffffull) { int r = trgl_flush(c); if (r) return r; }
    if (n > 0xffffffffull) return fail(c, TRGL_E_UNSUPPORTED, "trgl_draw: more than 2^32-1 triangles in one draw");

    const double* dclip = clip; const double* dvary = K ? vary : nullptr; const uint32_t* dcol = colors;
    struct Hold { trgl_ctx* c; Hold(trgl_ctx* x) : c(x) { ++c->stage_hold; } ~Hold() { --c->stage_hold; } } hold(c);
    if (mem_kind == TRGL_MEM_HOST) {
        void* p = nullptr; int r;
        if ((r = stage_copy(c, clip, n * 12 * sizeof(double), &p))) return r;
        dclip = (const double*)p;
        if (K) { if ((r = stage_copy(c, vary, n * K * sizeof(double), &p))) return r; dvary = (const double*)p; }
        if (colors) { if ((r = stage_copy(c, colors, n * sizeof(uint32_t), &p))) return r; dcol = (const uint32_t*)p; }
    }
    // a record addresses its triangle as (draw index, 24-bit index): split larger submissions
    for (uint64_t done = 0; done < n;) {
        uint64_t m = n - done; if (m > TRGL_DRAW_MAX_TRIS) m = TRGL_DRAW_MAX_TRIS;
        if (c->draws.size() >= TRGL_MAX_DRAWS || c->queued_tris + m > TRGL_FLUSH_MAX_TRIS) { int r = trgl_flush(c); if (r) return r; }
        DrawDesc d; std::memset(&d, 0, sizeof(d));
        d.n = (uint32_t)m; d.first = (uint32_t)c->queued_tris; d.kind = kind; d.K = K;
        if (u) d.u = *u; else { d.u.tex_diffuse = d.u.tex_normal = d.u.tex_specular = -1; }
        d.clip = dclip + done * 12;
        d.vary = dvary ? dvary + done * K : nullptr;
        d.colors = dcol ? dcol + done : nullptr;
        c->draws.push_back(d);
        c->queued_tris += m;
        done += m;
    }
    return TRGL_OK;
}

int trgl_draw_indexed(trgl_ctx* c, int kind, const trgl_uniforms* u, const double projection[16], const double* vertices,
                      int stride, uint64_t n_vertices, const uint32_t* indices, uint64_t n_faces, int mem_kind) {
    CHKCTX(c);
    if (c->rp.active) { int fr = trgl_flush_end(c); if (fr) return fr; }
    if (kind != TRGL_SHADER_PHONG && kind != TRGL_SHADER_EYE) return fail(c, TRGL_E_INVALID, "trgl_draw_indexed: kind must be PHONG or EYE");
    if (!u || !projection || !vertices || !indices) return fail(c, TRGL_E_INVALID, "trgl_draw_indexed: null argument");
    if (stride < 8) return fail(c, TRGL_E_INVALID, "trgl_draw_indexed: vertex stride must be >= 8 doubles (pos3, normal3, uv2)");
    if (mem_kind != TRGL_MEM_HOST && mem_kind != TRGL_MEM_DEVICE) return fail(c, TRGL_E_INVALID, "trgl_draw_indexed: bad mem_kind");
    if (n_faces == 0) return TRGL_OK;
    if (n_faces > 0xffffffffull / 3) return fail(c, TRGL_E_UNSUPPORTED, "trgl_draw_indexed: too many faces in one call");
    const double* dv = vertices; const uint32_t* di = indices;
    void* p = nullptr; int r;
    struct Hold { trgl_ctx* c; Hold(trgl_ctx* x) : c(x) { ++c->stage_hold; } ~Hold() { --c->stage_hold; } } hold(c);
    if (mem_kind == TRGL_MEM_HOST) {
        for (uint64_t k = 0; k < 3 * n_faces; ++k)
            if (indices[k] >= n_vertices) return fail(c, TRGL_E_INVALID, "trgl_draw_indexed: index out of range");
        if ((r = stage_copy(c, vertices, n_vertices * (size_t)stride * sizeof(double), &p))) return r;
        dv = (const double*)p;
        if ((r = stage_copy(c, indices, 3 * n_faces * sizeof(uint32_t), &p))) return r;
        di = (const uint32_t*)p;
    }
    double* clip = nullptr; double* vary = nullptr;
    if ((r = stage_alloc(c, n_faces * 12 * sizeof(double), &p))) return r;
    clip = (double*)p;
    if ((r = stage_alloc(c, n_faces * 24 * sizeof(double), &p))) return r;
    vary = (double*)p;
    launch_vertex_stage(c->stream, u->model_view, projection, dv, stride, di, (uint32_t)n_faces, clip, vary);
    HIPCHK(c, hipGetLastError());
    return trgl_draw(c, kind, u, clip, vary, nullptr, n_faces, TRGL_MEM_DEVICE);
}

void trgl_ssao_defaults(trgl_ssao_params* p) {      // main.cpp:317-321
    if (!p) return;
    p->num_directions = 8; p->steps_per_direction = 8; p->sample_radius = 16.0; p->occlusion_threshold = 1e-3; p->intensity = 0.35;
}

static int flush_sync(trgl_ctx* c);

int trgl_postprocess(trgl_ctx* c, const trgl_ssao_params* params, uint8_t* zimg, uint8_t* ao, uint8_t* fin) {
    CHKCTX(c);
    trgl_ssao_params sp; trgl_ssao_defaults(&sp);
    if (params) sp = *params;
    if (sp.num_directions < 1 || sp.num_directions > 16 || sp.steps_per_direction < 1)
        return fail(c, TRGL_E_INVALID, "trgl_postprocess: 1..16 directions, >= 1 step");
    int r = flush_sync(c); if (r) return r;
    const size_t npx = (size_t)c->W * c->H;
    // three [H][W][3] images, each at a 16-byte boundary (the kernels store dwords: W * H need not be a multiple of 4), then the keys
    const size_t img = (npx * 3 + 15) & ~size_t(15);
    if ((r = grow(c, c->pp_out, c->cap_pp, img * 3 + 64))) return r;      // allocated once per context, not per call
    uint8_t* d_out = c->pp_out;
    unsigned long long* d_keys = reinterpret_cast<unsigned long long*>(d_out + img * 3);
    uint8_t* d_z = d_out; uint8_t* d_ao = d_out + img; uint8_t* d_fin = d_out + img * 2;
    hipStream_t s = c->stream;
    if (zimg) launch_zimage(s, c->zb, c->W, c->H, d_keys, d_z);
    if (ao || fin) {
        double dx[16], dy[16];
        for (int d = 0; d < sp.num_directions; ++d) {         // main.cpp:333-334, host libm as in the reference
            double angle = 2.0 * 3.14159265358979323846 * d / sp.num_directions;
            dx[d] = std::cos(angle); dy[d] = std::sin(angle);
        }
        launch_ssao(s, c->zb, c->W, c->H, dx, dy, sp.num_directions, sp.steps_per_direction, sp.sample_radius,
                    sp.occlusion_threshold, sp.intensity, d_ao);
    }
    if (fin) {
        if (c->bpp < 3) return fail(c, TRGL_E_UNSUPPORTED, "trgl_postprocess: composite needs an RGB(A) framebuffer");
        launch_composite(s, c->fb, c->bpp, d_ao, c->W, c->H, d_fin);
    }
    HIPCHK(c, hipGetLastError());
    if (zimg) HIPCHK(c, hipMemcpyAsync(zimg, d_z, npx * 3, hipMemcpyDeviceToHost, s));
    if (ao) HIPCHK(c, hipMemcpyAsync(ao, d_ao, npx * 3, hipMemcpyDeviceToHost, s));
    if (fin) HIPCHK(c, hipMemcpyAsync(fin, d_fin, npx * 3, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return TRGL_OK;
}

static int resolve_events(trgl_ctx* c) {
    if (!c->events_pending) return TRGL_OK;
    HIPCHK(c, hipEventSynchronize(c->ev[3]));
    float ms;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); c->phase_ms[TRGL_PHASE_SETUP] += ms;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[1], c->ev[2])); c->phase_ms[TRGL_PHASE_BIN] += ms;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[2], c->ev[3])); c->phase_ms[TRGL_PHASE_RASTER] += ms;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[3])); c->phase_ms[TRGL_PHASE_TOTAL] += ms;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[4], c->ev[5])); c->phase_ms[TRGL_PHASE_RASTER_KERNEL] += ms;
    c->flushes_timed++;
    c->events_pending = false;
    return TRGL_OK;
}

int trgl_flush(trgl_ctx* c) {
    CHKCTX(c);
    int r = trgl_flush_begin(c);
    return r ? r : trgl_flush_end(c);
}

// expand -> stable radix passes by tile id -> per-tile bounds, for pair buffers of capacity `cap`.  All of it reads the pair
// count from device memory; when the count exceeds `cap` every kernel here does nothing.
static int queue_binning(trgl_ctx* c, const FrameParams& fp, uint32_t cap, int* cur_out) {
    hipStream_t s = c->stream;
    const size_t ntiles = (size_t)c->tiles_x * c->tiles_y;
    const unsigned long long* pairs_dev = &c->stats_dev->pairs_total;
    int r;
    uint32_t blk_base = 0;
    const bool key16 = ntiles <= 65536;        // tile indices as 16-bit keys: a third less traffic in every binning kernel
    for (auto& d : c->draws) {
        launch_expand(s, fp, d.first, d.n, c->tiles_x, c->cnt, c->blk_sums, c->chunk_off, blk_base, c->tilebox, c->keys[0], key16, c->vals[0], c->bmask[0], pairs_dev, cap);
        blk_base += setup_num_blocks(d.n);
    }
    int key_bits = 1; while ((size_t(1) << key_bits) < ntiles) ++key_bits;
    int passes = (key_bits + 7) / 8;
    int bits_per = (key_bits + passes - 1) / passes;
    size_t hist_need = ((size_t)radix_num_workers(cap) << bits_per) + 16;
    if ((r = grow(c, c->hist, c->cap_hist, hist_need))) return r;
    if ((r = grow(c, c->scan_tmp, c->cap_scan, 256 + 16))) return r;   // the digit totals of a pass (k_radix_scan_rows)
    int cur = 0;
    for (int ps = 0; ps < passes; ++ps) {
        launch_radix_pass(s, c->keys[cur], c->vals[cur], c->bmask[cur], c->keys[cur ^ 1], c->vals[cur ^ 1], c->bmask[cur ^ 1], key16, pairs_dev, cap, ps * bits_per, bits_per,
                          c->hist, c->scan_tmp);
        cur ^= 1;
    }
    launch_bounds(s, c->keys[cur], key16, pairs_dev, cap, c->tile_start, c->tile_end);
    *cur_out = cur;
    return TRGL_OK;
}

static int grow_pairs(trgl_ctx* c, size_t need) {
    if (need <= c->cap_pairs) return TRGL_OK;
    size_t ncap = (need + need / 4 + 1024 + 3) & ~size_t(3);     // a multiple of 4 entries: k_bounds reads 16 bytes at a time
    if (ncap > 0xffffe000ull) ncap = 0xffffe000ull;        // (grids are sized by cap + 4095 in 32 bits; a flush of 2^32 - 16 pairs and more is refused)
    int r;
    for (int k = 0; k < 2; ++k) {
        if ((r = realloc_dev(c, (void**)&c->keys[k], ncap * 4))) return r;
        if ((r = realloc_dev(c, (void**)&c->vals[k], ncap * 4))) return r;
        if ((r = realloc_dev(c, (void**)&c->bmask[k], ncap * 2))) return r;
    }
    c->cap_pairs = ncap;
    return TRGL_OK;
}

// First half of a flush: per-triangle setup and the stable tile binning.  Touches neither the framebuffer nor the
// z-buffer, so a caller may let it overlap with whatever still reads them (bench.py: the RCCL gather of the
// previous frame's strips).  The host does not wait for anything here: the number of (tile, triangle) pairs stays on the
// device, the binning kernels are queued for the capacity the pair buffers already have (an earlier flush's count + 25 %,
// or 2 pairs per triangle the first time), and trgl_flush_end checks the count - which has reached pinned memory long
// before the binning is through - and queues them again in the rare case that the buffers were too small.
int trgl_flush_begin(trgl_ctx* c) {
    CHKCTX(c);
    if (c->rp.active) return TRGL_OK;
    if (c->draws.empty() && !c->clear_pending) return TRGL_OK;
    int r;
    if ((r = resolve_events(c))) return r;
    const uint64_t N = c->queued_tris;
    const size_t ntiles = (size_t)c->tiles_x * c->tiles_y;
    hipStream_t s = c->stream;

    FrameParams fp; std::memset(&fp, 0, sizeof(fp));
    fp.fb = c->fb; fp.zb = c->zb; fp.W = c->W; fp.H = c->H; fp.bpp = c->bpp;
    fp.tiles_x = c->tiles_x; fp.tiles_y = c->tiles_y;
    fp.strip_y0 = c->strip_y0; fp.strip_y1 = c->strip_y1;
    fp.strip_ty0 = c->strip_y0 / TRGL_TILE;
    fp.strip_ty1 = (c->strip_y1 + TRGL_TILE - 1) / TRGL_TILE;
    if (c->strip_y1 <= c->strip_y0) fp.strip_ty1 = fp.strip_ty0;
    fp.il_tiles = c->il_tiles; fp.il_world = c->il_world; fp.il_rank = c->il_rank;
    fp.init_from_clear = c->clear_pending ? 1 : 0;
    fp.n_tris = (uint32_t)c->queued_tris;
    fp.clear_color = c->clear_color; fp.clear_z = c->clear_z;
    std::memcpy(fp.vp, c->vp, sizeof(fp.vp));

    int flush_kind = c->draws.empty() ? TRGL_SHADER_FLAT : c->draws[0].kind;     // one kind for the whole flush, or -1
    for (auto& d : c->draws) if (d.kind != flush_kind) flush_kind = -1;
    bool shade_later = false;
    for (auto& d : c->draws) if (d.kind == TRGL_SHADER_PHONG || d.kind == TRGL_SHADER_EYE) shade_later = true;
    if (shade_later) {                                                            // shaded once per visible pixel (k_shade)
        if ((r = grow(c, c->idbuf, c->cap_idbuf, (size_t)c->W * c->H))) return r;
        fp.idbuf = c->idbuf;
    }

    if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[0], s));
    uint32_t cap = 0;
    int cur = 0;
    if (N) {
        // grow per-triangle buffers together
        if (N + 1 > c->cap_tris) {                     // (+ 1: the record behind the last one, k_make_items)
            size_t ncap = N + N / 4 + 1024;
            if ((r = realloc_dev(c, (void**)&c->recs, ncap * sizeof(TriRec)))) return r;
            if ((r = realloc_dev(c, (void**)&c->recs_w, ncap * sizeof(TriW)))) return r;
            if ((r = realloc_dev(c, (void**)&c->cnt, ncap * 4))) return r;
            if ((r = realloc_dev(c, (void**)&c->tilebox, ncap * sizeof(uint2)))) return r;
            c->cap_tris = ncap;
        }
        if (c->cap_pairs == 0 && (r = grow_pairs(c, (size_t)2 * N + 4096))) return r;      // first flush: a guess, checked in trgl_flush_end
        uint32_t nblk = 0;
        for (auto& d : c->draws) nblk += setup_num_blocks(d.n);
        if ((r = grow(c, c->blk_sums, c->cap_blk, (size_t)nblk + 16))) return r;
        if ((r = grow(c, c->chunk_off, c->cap_chunk, (size_t)nblk / 16 + 16))) return r;
        {
            uint32_t blk_base = 0;
            for (size_t i = 0; i < c->draws.size(); ++i) {
                launch_setup(s, fp, c->draws[i], c->draws_dev, (int)i, c->draws[i].n, c->recs, c->recs_w, c->cnt, c->tilebox, c->stats_dev, c->blk_sums, blk_base);
                blk_base += setup_num_blocks(c->draws[i].n);
            }
        }
        if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[1], s));
        // (+ literal_tris, large_tris into pinned memory; the kernel also clears tile_start and tile_end)
        launch_chunk_spine(s, c->blk_sums, nblk, c->chunk_off, &c->stats_dev->pairs_total, &c->stats_pinned->pairs_total,
                           c->tile_start, (ntiles * 8 + 15) & ~size_t(15));
        HIPCHK(c, hipEventRecord(c->ev_pairs, s));
        cap = (uint32_t)c->cap_pairs;
        if ((r = queue_binning(c, fp, cap, &cur))) return r;
    } else {
        if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[1], s));
        HIPCHK(c, hipMemsetAsync(c->tile_start, 0, ntiles * 8, s));
    }
    if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[2], s));
    c->rp.active = true; c->rp.fp = fp; c->rp.flush_kind = flush_kind; c->rp.cap = cap; c->rp.cur = cur; c->rp.N = N;
    return TRGL_OK;
}

// Second half: the tile raster (and k_shade), the only part that reads or writes the framebuffer and the z-buffer.
int trgl_flush_end(trgl_ctx* c) {
    CHKCTX(c);
    if (!c->rp.active) return TRGL_OK;
    c->rp.active = false;
    int r;
    hipStream_t s = c->stream;
    FrameParams fp = c->rp.fp;
    const int flush_kind = c->rp.flush_kind;
    int cur = c->rp.cur;
    const uint64_t N = c->rp.N;
    uint32_t P = 0;
    if (N) {
        // the pair count was copied to pinned memory right after k_setup; the GPU is busy with the binning queued behind it
        HIPCHK(c, hipEventSynchronize(c->ev_pairs));
        const unsigned long long P64 = c->stats_pinned->pairs_total;
        if (P64 > 0xffffe000ull) {
            c->draws.clear(); c->queued_tris = 0;          // nothing of this flush was drawn (every binning kernel saw the overflow)
            return fail(c, TRGL_E_UNSUPPORTED, "flush: more than 2^32 triangle-tile pairs; submit in smaller batches");
        }
        P = (uint32_t)P64;
        if (P > c->rp.cap) {                               // the buffers were too small: the queued binning did nothing
            if ((r = grow_pairs(c, P))) return r;
            if ((r = queue_binning(c, fp, (uint32_t)c->cap_pairs, &cur))) return r;
            if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[2], s));
        }
    }
    // with no pairs every tile list is empty; the kernel must not (and does not) dereference these, but give it
    // valid addresses anyway
    const TriRec* recs_arg = c->recs ? c->recs : reinterpret_cast<const TriRec*>(c->tile_start);
    const uint32_t* vals_arg = (P && c->vals[cur]) ? c->vals[cur] : c->tile_start;
    const uint16_t* bmask_arg = (P && c->bmask[cur]) ? c->bmask[cur] : reinterpret_cast<const uint16_t*>(c->tile_start);
    // one work item (a workgroup of four block waves) per row of blocks of every owned tile
    const uint32_t max_items = raster_max_items(fp);
    if ((r = grow(c, c->items, c->cap_items, (size_t)max_items + 64))) return r;
    if ((r = grow(c, c->item_stats, c->cap_item_stats, ((size_t)max_items + 64) * 4))) return r;
    // k_setup counted the triangles that are not well scaled (it came over with the pair count): without any, the kernel without the literal path
    const bool all_well_scaled = N == 0 || c->stats_pinned->literal_tris == 0;
    fp.zq_cull = (N != 0 && c->stats_pinned->large_tris != 0) ? 1 : 0;      // (k_setup counted them; the count came over with the pair count)
    if (std::getenv("TRGL_DEBUG_PTRS")) {        // diagnostics: where the buffers of this flush live
        std::fprintf(stderr, "trgl ptrs: fb %p +%zu  zb %p +%zu  recs %p +%zu  recs_w %p  vals %p bmask %p cap_pairs %zu P %u  items %p cap %zu  item_stats %p  tile_start %p  N %llu max_items %u\n",
                     (void*)c->fb, (size_t)c->W * c->H * c->bpp, (void*)c->zb, (size_t)c->W * c->H * 8, (const void*)recs_arg, c->cap_tris * sizeof(TriRec), (void*)c->recs_w,
                     (const void*)vals_arg, (const void*)bmask_arg, c->cap_pairs, P, (void*)c->items, c->cap_items, (void*)c->item_stats, (void*)c->tile_start, (unsigned long long)N, max_items);
    }
    launch_raster(s, fp, flush_kind, all_well_scaled, recs_arg, c->recs_w, vals_arg, bmask_arg, c->tile_start, c->tile_end, c->draws_dev, c->tex_dev, c->stats_dev,
                  max_items, c->items, c->n_items, c->item_stats,
                  c->profiling ? c->ev[4] : nullptr, c->profiling ? c->ev[5] : nullptr);
    if (c->profiling) { HIPCHK(c, hipEventRecord(c->ev[3], s)); c->events_pending = true; }
    HIPCHK(c, hipGetLastError());

    c->triangles_total += N;
    c->last_tris = N; c->last_pairs = P;
    c->clear_pending = false;
    bool had_stage = false;
    for (auto& ch : c->stage) if (ch.used) had_stage = true;
    c->draws.clear();
    c->queued_tris = 0;
    if (had_stage) {                       // staged data may be recycled only once the kernels are done ...
        HIPCHK(c, hipStreamSynchronize(s));
        if (!c->stage_hold)                // ... and not while a draw call in progress still owns staged arrays
            for (auto& ch : c->stage) ch.used = 0;
    }
    return TRGL_OK;
}

int trgl_sync(trgl_ctx* c) {
    CHKCTX(c);
    if (c->rp.active) { int fr = trgl_flush_end(c); if (fr) return fr; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return TRGL_OK;
}

static int flush_sync(trgl_ctx* c) {
    int r = trgl_flush(c); if (r) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return TRGL_OK;
}

int trgl_read_framebuffer(trgl_ctx* c, uint8_t* dst) {
    CHKCTX(c);
    if (!dst) return fail(c, TRGL_E_INVALID, "null destination");
    int r = flush_sync(c); if (r) return r;
    HIPCHK(c, hipMemcpy(dst, c->fb, (size_t)c->W * c->H * c->bpp, hipMemcpyDeviceToHost));
    return TRGL_OK;
}
int trgl_write_framebuffer(trgl_ctx* c, const uint8_t* src) {
    CHKCTX(c);
    if (!src) return fail(c, TRGL_E_INVALID, "null source");
    int r = flush_sync(c); if (r) return r;
    HIPCHK(c, hipMemcpy(c->fb, src, (size_t)c->W * c->H * c->bpp, hipMemcpyHostToDevice));
    return TRGL_OK;
}
int trgl_read_zbuffer(trgl_ctx* c, double* dst) {
    CHKCTX(c);
    if (!dst) return fail(c, TRGL_E_INVALID, "null destination");
    int r = flush_sync(c); if (r) return r;
    HIPCHK(c, hipMemcpy(dst, c->zb, (size_t)c->W * c->H * sizeof(double), hipMemcpyDeviceToHost));
    return TRGL_OK;
}
int trgl_write_zbuffer(trgl_ctx* c, const double* src) {
    CHKCTX(c);
    if (!src) return fail(c, TRGL_E_INVALID, "null source");
    int r = flush_sync(c); if (r) return r;
    HIPCHK(c, hipMemcpy(c->zb, src, (size_t)c->W * c->H * sizeof(double), hipMemcpyHostToDevice));
    return TRGL_OK;
}

int trgl_get_stats(trgl_ctx* c, trgl_stats* out) {
    CHKCTX(c);
    if (!out) return fail(c, TRGL_E_INVALID, "null stats");
    int r = flush_sync(c); if (r) return r;
    HIPCHK(c, hipMemcpy(c->stats_pinned, c->stats_dev, sizeof(DevStats), hipMemcpyDeviceToHost));
    const DevStats& s = *c->stats_pinned;
    out->triangles_rasterized = c->triangles_total;
    out->fragments_drawn = s.fragments;
    out->min_x = s.min_x; out->min_y = s.min_y; out->max_x = s.max_x; out->max_y = s.max_y;
    out->min_z = zkey_decode(s.zmin_key); out->max_z = zkey_decode(s.zmax_key);
    // +0.0 == -0.0: the reference keeps whichever zero it met first (std::min/max, our_gl.cpp:197-198)
    if (out->min_z == 0.0 && s.zero_locked) out->min_z = s.zero_sign ? -0.0 : 0.0;
    if (out->max_z == 0.0 && s.zero_locked) out->max_z = s.zero_sign ? -0.0 : 0.0;
    return TRGL_OK;
}

int trgl_reset_stats(trgl_ctx* c) {
    CHKCTX(c);
    int r = flush_sync(c); if (r) return r;
    c->triangles_total = 0;
    return reset_dev_stats(c);
}

int trgl_format_stats(const trgl_stats* s, char* buf, size_t buflen) {   // our_gl.cpp:205-209
    if (!s || !buf) return TRGL_E_INVALID;
    char lo[400], hi[400];      // "%f" of a double needs up to 317 characters
    if (std::isfinite(s->min_z)) std::snprintf(lo, sizeof lo, "%f", s->min_z); else std::snprintf(lo, sizeof lo, "inf");
    if (std::isfinite(s->max_z)) std::snprintf(hi, sizeof hi, "%f", s->max_z); else std::snprintf(hi, sizeof hi, "-inf");
    int n = std::snprintf(buf, buflen, "DEBUG: triangles=%llu fragments_drawn=%llu bbox=[%d,%d] - [%d,%d] z-range=[%s,%s]\n",
                          (unsigned long long)s->triangles_rasterized, (unsigned long long)s->fragments_drawn,
                          s->min_x, s->min_y, s->max_x, s->max_y, lo, hi);
    return (n < 0 || (size_t)n >= buflen) ? TRGL_E_INVALID : TRGL_OK;
}

void* trgl_framebuffer_device_ptr(trgl_ctx* c) { return c ? c->fb : nullptr; }
void* trgl_zbuffer_device_ptr(trgl_ctx* c) { return c ? c->zb : nullptr; }
void* trgl_stream(trgl_ctx* c) { return c ? (void*)c->stream : nullptr; }

int trgl_obj_load(const char* path, double** vertices, uint64_t* n_vertices, uint32_t** indices, uint64_t* n_faces) {
    if (!path || !vertices || !n_vertices || !indices || !n_faces) return TRGL_E_INVALID;
    trgl_obj::Mesh m;
    try {
        if (!trgl_obj::load(path, m)) { g_create_error = m.error; return TRGL_E_INVALID; }
    } catch (const std::bad_alloc&) {
        g_create_error = "trgl_obj_load: out of memory"; return TRGL_E_NOMEM;
    }
    *n_vertices = m.vertices.size() / 14; *n_faces = m.indices.size() / 3;
    *vertices = (double*)std::malloc(m.vertices.size() * sizeof(double) + 8);
    *indices = (uint32_t*)std::malloc(m.indices.size() * sizeof(uint32_t) + 8);
    if (!*vertices || !*indices) { std::free(*vertices); std::free(*indices); return TRGL_E_NOMEM; }
    std::memcpy(*vertices, m.vertices.data(), m.vertices.size() * sizeof(double));
    std::memcpy(*indices, m.indices.data(), m.indices.size() * sizeof(uint32_t));
    return TRGL_OK;
}
void trgl_obj_free(double* vertices, uint32_t* indices) { std::free(vertices); std::free(indices); }

size_t trgl_tga_max_size(int w, int h, int bpp) {
    if (w <= 0 || h <= 0 || bpp <= 0) return 18;
    return size_t(18) + size_t(w) * h * bpp + size_t(w) * h;      // every pixel its own literal packet
}

int trgl_tga_encode(const uint8_t* pixels, int w, int h, int bpp, int vflip, int rle, uint8_t* out, size_t* out_len) {
    if (!pixels || !out || !out_len || w <= 0 || h <= 0 || w > 65535 || h > 65535 || !(bpp == 1 || bpp == 3 || bpp == 4)) return TRGL_E_INVALID;
    try {
        TGAImage img(w, h, bpp);
        std::memcpy(img.buffer(), pixels, size_t(w) * h * bpp);
        std::vector<uint8_t> bytes = img.encode_tga(vflip != 0, rle != 0);
        std::memcpy(out, bytes.data(), bytes.size());
        *out_len = bytes.size();
    } catch (const std::bad_alloc&) {
        return TRGL_E_NOMEM;
    }
    return TRGL_OK;
}

int trgl_tga_info(const uint8_t* file, size_t size, int* width, int* height, int* bpp) {
    if (!file || !width || !height || !bpp || size < 18) return TRGL_E_INVALID;            // tgaimage.cpp:85-90
    const int w = file[12] | (file[13] << 8), h = file[14] | (file[15] << 8), b = file[16] >> 3;
    if (w <= 0 || h <= 0 || (b != 1 && b != 3 && b != 4)) return TRGL_E_INVALID;           // :96-99
    if (!(file[2] == 2 || file[2] == 3 || file[2] == 10 || file[2] == 11)) return TRGL_E_INVALID;   // :113-116
    *width = w; *height = h; *bpp = b;
    return TRGL_OK;
}

int trgl_tga_decode(const uint8_t* file, size_t size, uint8_t* pixels) {
    if (!file || !pixels) return TRGL_E_INVALID;
    try {                                          // a header may claim 65535 x 65535 x 4 bytes: nothing throws across the C ABI
        TGAImage img;
        if (!img.decode_tga(file, size)) return TRGL_E_INVALID;
        std::memcpy(pixels, img.buffer(), size_t(img.width()) * img.height() * img.bytespp());
    } catch (const std::bad_alloc&) {
        return TRGL_E_NOMEM;
    }
    return TRGL_OK;
}

int trgl_selftest_division(trgl_ctx* c, uint64_t samples, uint64_t seed, uint64_t* mismatches) {
    CHKCTX(c);
    if (!mismatches) return fail(c, TRGL_E_INVALID, "null mismatches");
    int r = trgl_flush(c); if (r) return r;
    unsigned long long* d = nullptr;
    HIPCHK(c, hipMalloc((void**)&d, 8));
    HIPCHK(c, hipMemsetAsync(d, 0, 8, c->stream));
    unsigned long long per_thread = (samples + 1024ull * 256 - 1) / (1024ull * 256);
    launch_selftest_division(c->stream, per_thread, seed, d);
    unsigned long long h = 0;
    HIPCHK(c, hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(d));
    *mismatches = h;
    return TRGL_OK;
}

int trgl_selftest_sampler(trgl_ctx* c, int slot, const double* uv, uint64_t n, uint8_t* out) {
    CHKCTX(c);
    if (!uv || !out) return fail(c, TRGL_E_INVALID, "trgl_selftest_sampler: null argument");
    if (slot < 0 || slot >= TRGL_MAX_TEXTURES) return fail(c, TRGL_E_INVALID, "trgl_selftest_sampler: bad slot");
    int r = trgl_flush(c); if (r) return r;
    if (!n) return TRGL_OK;
    double* d_uv = nullptr; uint8_t* d_out = nullptr;
    HIPCHK(c, hipMalloc((void**)&d_uv, n * 16));
    HIPCHK(c, hipMalloc((void**)&d_out, n * 5));
    HIPCHK(c, hipMemcpyAsync(d_uv, uv, n * 16, hipMemcpyHostToDevice, c->stream));
    launch_selftest_sampler(c->stream, c->tex_dev, slot, d_uv, n, d_out);
    HIPCHK(c, hipMemcpyAsync(out, d_out, n * 5, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(d_uv)); HIPCHK(c, hipFree(d_out));
    return TRGL_OK;
}

int trgl_set_stream(trgl_ctx* c, void* hip_stream, int use_own) {
    CHKCTX(c);
    if (c->rp.active) { int fr = trgl_flush_end(c); if (fr) return fr; }
    int r = TRGL_OK;
    if (!c->draws.empty()) r = trgl_flush(c);      // a pending clear alone needs no launch: it stays pending
    if (r) return r;
    if ((r = resolve_events(c))) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // a NULL hipStream_t is a real stream (the legacy default stream, which is what torch's current stream
    // usually is), so "go back to the context's own stream" is a separate flag
    c->stream = use_own ? c->own_stream : (hipStream_t)hip_stream;
    return TRGL_OK;
}

int trgl_set_profiling(trgl_ctx* c, int on) {
    CHKCTX(c);
    int r = resolve_events(c); if (r) return r;
    c->profiling = on != 0;
    return TRGL_OK;
}
int trgl_get_phase_ms(trgl_ctx* c, double ms[TRGL_NUM_PHASES], uint64_t* flushes) {
    CHKCTX(c);
    int r = resolve_events(c); if (r) return r;
    if (ms) for (int i = 0; i < TRGL_NUM_PHASES; ++i) ms[i] = c->phase_ms[i];
    if (flushes) *flushes = c->flushes_timed;
    return TRGL_OK;
}
int trgl_reset_phase_ms(trgl_ctx* c) {
    CHKCTX(c);
    int r = resolve_events(c); if (r) return r;
    for (int i = 0; i < TRGL_NUM_PHASES; ++i) c->phase_ms[i] = 0;
    c->flushes_timed = 0;
    return TRGL_OK;
}
// diagnostic builds (-DTRGL_DEBUG_COUNTERS): k_raster work counters since the last stats reset
extern "C" int trgl_debug_counters(trgl_ctx* c, unsigned long long out[16]) {
    CHKCTX(c);
    int r = flush_sync(c); if (r) return r;
    HIPCHK(c, hipMemcpy(c->stats_pinned, c->stats_dev, sizeof(DevStats), hipMemcpyDeviceToHost));
    for (int k = 0; k < 16; ++k) out[k] = c->stats_pinned->dbg[k];
    return TRGL_OK;
}

}  // extern "C"

// ---- RCCL, loaded on demand (the library has no link-time dependency on it) --------------------------------------------
namespace {
struct Rccl {
    struct Id { char b[128]; };          // ncclUniqueId, passed by value (rccl.h)
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl r;
    if (!r.lib) {
        r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) {
            r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
            r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
            r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
            r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
            r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
            r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
            r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
            r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd;
        }
    }
    return r;
}
std::string rccl_err(const char* what, int code) {
    Rccl& r = rccl();
    return std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(code) : "RCCL error") + " (" + std::to_string(code) + ")";
}
constexpr int NCCL_UINT8 = 1;       // ncclUint8 (rccl.h, ncclDataType_t)
}  // namespace

extern "C" {

int trgl_rccl_unique_id(uint8_t id[TRGL_RCCL_ID_BYTES]) {
    if (!id) return TRGL_E_INVALID;
    Rccl& r = rccl();
    if (!r.ok) { g_create_error = "librccl.so.1 could not be loaded"; return TRGL_E_UNSUPPORTED; }
    const int rc = r.GetUniqueId(id);
    if (rc) { g_create_error = rccl_err("ncclGetUniqueId", rc); return TRGL_E_HIP; }
    return TRGL_OK;
}
int trgl_rccl_comm_create(const uint8_t id[TRGL_RCCL_ID_BYTES], int rank, int world, int device, void** comm) {
    if (!id || !comm || world < 1 || rank < 0 || rank >= world) return TRGL_E_INVALID;
    Rccl& r = rccl();
    if (!r.ok) { g_create_error = "librccl.so.1 could not be loaded"; return TRGL_E_UNSUPPORTED; }
    if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return TRGL_E_HIP; }
    Rccl::Id uid; std::memcpy(uid.b, id, 128);
    const int rc = r.CommInitRank(comm, world, uid, rank);
    if (rc) { g_create_error = rccl_err("ncclCommInitRank", rc); return TRGL_E_HIP; }
    return TRGL_OK;
}
int trgl_rccl_comm_destroy(void* comm) {
    if (!comm) return TRGL_E_INVALID;
    Rccl& r = rccl();
    if (!r.ok) return TRGL_E_UNSUPPORTED;
    return r.CommDestroy(comm) ? TRGL_E_HIP : TRGL_OK;
}

int trgl_gather(trgl_ctx* c, void* comm, int rank, int world, int with_z) {
    CHKCTX(c);
    if (!comm || world < 1 || rank < 0 || rank >= world) return fail(c, TRGL_E_INVALID, "trgl_gather: bad communicator / rank / world");
    Rccl& r = rccl();
    if (!r.ok) return fail(c, TRGL_E_UNSUPPORTED, "trgl_gather: librccl.so.1 could not be loaded");
    int fr = trgl_flush(c); if (fr) return fr;               // the rows this context owns are complete behind this point of the stream
    const size_t row_fb = (size_t)c->W * c->bpp, row_z = (size_t)c->W * sizeof(double);
    int rc = 0;
    if (c->il_tiles == 0) {
        // one strip per rank: equal, contiguous chunks of the row-major buffers
        if (c->H % world) return fail(c, TRGL_E_INVALID, "trgl_gather: the height is not divisible by the number of ranks (equal strips are required)");
        const int rows = c->H / world;
        if (c->strip_y0 != rank * rows || c->strip_y1 != (rank + 1) * rows)
            return fail(c, TRGL_E_STATE, "trgl_gather: this context's strip is not rows [rank * H / world, (rank + 1) * H / world)");
        if ((rc = r.GroupStart())) { c->err = rccl_err("ncclGroupStart", rc); return TRGL_E_HIP; }
        rc = r.AllGather(c->fb + (size_t)c->strip_y0 * row_fb, c->fb, (size_t)rows * row_fb, NCCL_UINT8, comm, c->stream);
        if (!rc && with_z) rc = r.AllGather(reinterpret_cast<uint8_t*>(c->zb) + (size_t)c->strip_y0 * row_z, c->zb, (size_t)rows * row_z, NCCL_UINT8, comm, c->stream);
        const int rc2 = r.GroupEnd();
        if (!rc) rc = rc2;
    } else {
        // interleaved bands: inside each period of world * band_rows rows the bands lie in rank order
        if (c->il_world != world || c->il_rank != rank) return fail(c, TRGL_E_STATE, "trgl_gather: rank / world differ from trgl_set_interleave");
        const int band = c->il_tiles * TRGL_TILE, period = band * world;
        if (c->H % period) return fail(c, TRGL_E_INVALID, "trgl_gather: the height is not a multiple of world * band_rows");
        if ((rc = r.GroupStart())) { c->err = rccl_err("ncclGroupStart", rc); return TRGL_E_HIP; }
        for (int p0 = 0; p0 < c->H && !rc; p0 += period) {
            const int y0 = p0 + rank * band;
            rc = r.AllGather(c->fb + (size_t)y0 * row_fb, c->fb + (size_t)p0 * row_fb, (size_t)band * row_fb, NCCL_UINT8, comm, c->stream);
            if (!rc && with_z) rc = r.AllGather(reinterpret_cast<uint8_t*>(c->zb) + (size_t)y0 * row_z, reinterpret_cast<uint8_t*>(c->zb) + (size_t)p0 * row_z,
                                                (size_t)band * row_z, NCCL_UINT8, comm, c->stream);
        }
        const int rc2 = r.GroupEnd();
        if (!rc) rc = rc2;
    }
    if (rc) { c->err = rccl_err("ncclAllGather", rc); return TRGL_E_HIP; }
    return TRGL_OK;
}

int trgl_get_last_flush_info(trgl_ctx* c, uint64_t* triangles, uint64_t* pairs, uint64_t* tiles) {
    if (!c) return TRGL_E_INVALID;
    if (triangles) *triangles = c->last_tris;
    if (pairs) *pairs = c->last_pairs;
    if (tiles) *tiles = (uint64_t)c->tiles_x * c->tiles_y;
    return TRGL_OK;
}

}  // extern "C"
