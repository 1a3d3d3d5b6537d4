// trgl_gl.h — the reference's our_gl.h surface (our_gl.h:17-61) over the C ABI of include/trgl.h.
//
// Same names, argument meaning and (silent) error behaviour as the reference:
//   globals  ModelView / Perspective / Viewport / zbuffer        (our_gl.h:17-20)
//   lookat, init_perspective, init_viewport, init_zbuffer        (our_gl.h:25-31)
//   struct IShader, typedef Triangle, rasterize(), print_render_stats()  (our_gl.h:36-61)
// What changes for a caller, and why:
//   * rasterize() is DEFERRED: it snapshots the clip coordinates, the shader's varyings and the uniforms that can
//     change between calls (the global ModelView is read inside fragment(), main.cpp:116) and batches them; the GPU
//     runs them in submission order.  `zbuffer` is a proxy whose accessors complete the pending work first, so the
//     reference's direct uses of it (main.cpp:700,730,751,759) need no edit; the framebuffer is the caller's own TGAImage,
//     so gl_flush(framebuffer) goes before the places that read its pixels (main.cpp:743,773).
//   * gl_draw_model(model, shader, framebuffer) replaces a whole face loop (main.cpp:660-666,692-698,715-721): the
//     vertex stage runs on the device from Model::vertices / indices (trgl_draw_indexed), 112 B per vertex + 12 B per face
//     cross PCIe instead of 288 + 96 B per face.
//   * a C++ virtual cannot be called from a kernel: IShader gains describe(), returning the POD descriptor of a
//     shader kind the device implements (trgl_shaders.h: FlatShader, GouraudShader, PhongShader, EyeShader).
//     A subclass without one makes rasterize() fail loudly — there is NO CPU fallback.
//   * errors of the C ABI (out of memory, a flush beyond 2^32 triangle-tile pairs, a HIP error ...) do not end the process: the
//     call that met one drops its work, gl_flush() / gl_draw_model() / gl_draw_indexed() / gl_postprocess() return false, and
//     gl_last_error() / gl_last_error_message() tell which (sticky until gl_clear_error()).  Only a programming error - an
//     IShader without describe() - still aborts.
// Header-only (C++17 inline variables); link with -ltrgl.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include <cstring>

#include "../../include/trgl.h"
// the caller's own value types (the reference's geometry.h) or the repo's minimal ones
#ifdef TRGL_GEOMETRY_HEADER
#include TRGL_GEOMETRY_HEADER
#else
#include "trgl_geometry.h"
#endif
#ifdef TRGL_IMAGE_HEADER
#include TRGL_IMAGE_HEADER
#else
#include "trgl_image.h"
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

inline mat<4, 4> ModelView = mat<4, 4>::identity();
inline mat<4, 4> Perspective = mat<4, 4>::identity();
inline mat<4, 4> Viewport = mat<4, 4>::identity();

// The reference's global `std::vector<double> zbuffer` (our_gl.h:20) with the depths living in HBM: every accessor first
// completes what rasterize() has batched and brings the depths to the host copy; anything that can modify the host copy
// marks it, and the next draw uploads it.  `std::vector<double> saved = zbuffer;` (main.cpp:700), `zbuffer = saved;`
// (main.cpp:730), `zbuffer[idx]`, `zbuffer.size()` and passing it as `const std::vector<double>&` (main.cpp:751,759)
// compile and behave as with the vector.
class trgl_zbuffer_proxy {
    std::vector<double> host_;
    void pull() const;                       // pending draws -> device -> host_
    void touched();                          // the caller may have written host_
public:
    operator const std::vector<double>&() const { pull(); return host_; }
    trgl_zbuffer_proxy& operator=(const std::vector<double>& v) { pull(); host_ = v; touched(); return *this; }
    double operator[](std::size_t i) const { pull(); return host_[i]; }
    // `zbuffer[i]` on the (non-const) global: reading it must not mark the host copy as modified - that would upload all W * H
    // depths again before the next draw - so the element comes back as a small reference object that marks on ASSIGNMENT only
    class element {
        trgl_zbuffer_proxy& z_; std::size_t i_;
    public:
        element(trgl_zbuffer_proxy& z, std::size_t i) : z_(z), i_(i) {}
        operator double() const { z_.pull(); return z_.host_[i_]; }
        element& operator=(double v) { z_.pull(); z_.host_[i_] = v; z_.touched(); return *this; }
        element& operator=(const element& o) { return *this = double(o); }
    };
    element operator[](std::size_t i) { return element(*this, i); }
    std::size_t size() const { return host_.size(); }
    bool empty() const { return host_.empty(); }
    const double* data() const { pull(); return host_.data(); }
    double* data() { pull(); touched(); return host_.data(); }
    std::vector<double>::const_iterator begin() const { pull(); return host_.begin(); }
    std::vector<double>::const_iterator end() const { pull(); return host_.end(); }     // (begin() and end() may be evaluated in either order)
    void assign(std::size_t n, double v);    // init_zbuffer (our_gl.cpp:72-74)
    void resize(std::size_t n) { host_.resize(n); }
    std::vector<double>& raw() { return host_; }   // the shim's own access: no synchronisation
};
inline trgl_zbuffer_proxy zbuffer;

// What a shader hands to the device: kind + uniforms + this triangle's varyings/colour (include/trgl.h).
struct trgl_shader_desc {
    int kind = TRGL_SHADER_FLAT;
    trgl_uniforms uniforms{};
    const double* varyings = nullptr;   // K doubles for `kind`, valid until rasterize() returns
    std::uint32_t color = 0xffffffffu;  // FLAT / GOURAUD
};

struct IShader {
    static TGAColor sample2D(const TGAImage& img, const vec2& uv) {
        int x = std::min<int>(img.width() - 1, std::max<int>(0, int(uv.x * img.width())));
        int y = std::min<int>(img.height() - 1, std::max<int>(0, int(uv.y * img.height())));
        return img.get(x, y);
    }
    virtual vec4 vertex(int, int) { return vec4(); }
    // Executed on the device; never called on the host.
    virtual std::pair<bool, TGAColor> fragment(const vec3) const {
        std::fprintf(stderr, "trgl: IShader::fragment() runs on the GPU; give the shader a describe()\n");
        std::abort();
    }
    virtual bool describe(trgl_shader_desc&) const { return false; }
    virtual ~IShader() = default;
};

typedef vec<4> Triangle[3];

namespace trgl_shim {

// The image type is the caller's (the reference's TGAImage keeps its bytes-per-pixel private and buffer() non-const,
// tgaimage.h:67-104): everything the shim needs goes through these three.
template <class Img> inline int image_bpp(const Img& im) { return (im.width() > 0 && im.height() > 0) ? int(im.get(0, 0).bytespp) : 0; }   // TGAColor(p, bpp), tgaimage.h:46-50
template <class Img> inline std::uint8_t* image_bytes(const Img& im) { return const_cast<Img&>(im).buffer(); }
inline std::uint32_t pack_bgra(const TGAColor& c) {
    return std::uint32_t(c.bgra[0]) | (std::uint32_t(c.bgra[1]) << 8) | (std::uint32_t(c.bgra[2]) << 16) | (std::uint32_t(c.bgra[3]) << 24);
}

struct State {
    trgl_ctx* ctx = nullptr;
    int w = 0, h = 0, bpp = 0;
    bool zbuffer_dirty_on_host = false;   // init_zbuffer()/host writes not yet on the device
    bool zbuffer_stale_on_host = false;   // the device drew since the host copy was fetched
    bool have_batch = false;
    int kind = 0;
    trgl_uniforms uniforms{};
    std::vector<double> clip, vary;
    std::vector<std::uint32_t> colors;
    mat<4, 4> viewport_at_batch;
    int err = TRGL_OK;                    // first C-ABI error since gl_clear_error() (a TRGL_E_* code)
    std::string err_msg;
};
inline State& state() { static State s; return s; }

// a C-ABI call failed: remember the first error (code + message); the caller of the shim asks gl_last_error()
inline bool fail(const char* what, int code, trgl_ctx* c) {
    State& s = state();
    if (s.err == TRGL_OK) { s.err = code; s.err_msg = std::string(what) + ": " + trgl_last_error(c); }
    return false;
}
// evaluates to true when the call succeeded
#define TRGL_SHIM_OK(call) ([&]() -> bool { const int rc_ = (call); return rc_ == TRGL_OK ? true : ::trgl_shim::fail(#call, rc_, ::trgl_shim::state().ctx); }())

inline int device_from_env() { const char* e = std::getenv("TRGL_DEVICE"); return e ? std::atoi(e) : 0; }

// make sure a context matching the framebuffer exists and holds the host's current pixels / depths
inline bool bind(TGAImage& fb) {
    State& s = state();
    const int fb_bpp = image_bpp(fb);
    if (s.ctx && (s.w != fb.width() || s.h != fb.height() || s.bpp != fb_bpp)) { trgl_destroy(s.ctx); s.ctx = nullptr; }
    if (!s.ctx) {
        const int rc = trgl_create(device_from_env(), fb.width(), fb.height(), fb_bpp, &s.ctx);
        if (rc != TRGL_OK) { s.ctx = nullptr; return fail("trgl_create", rc, nullptr); }
        s.w = fb.width(); s.h = fb.height(); s.bpp = fb_bpp;
        if (!TRGL_SHIM_OK(trgl_write_framebuffer(s.ctx, fb.buffer()))) return false;
        s.zbuffer_dirty_on_host = true;
    }
    if (s.zbuffer_dirty_on_host) {
        std::vector<double>& hz = zbuffer.raw();
        if (hz.size() != std::size_t(s.w) * s.h) hz.assign(std::size_t(s.w) * s.h, std::numeric_limits<double>::infinity());
        if (!TRGL_SHIM_OK(trgl_write_zbuffer(s.ctx, hz.data()))) return false;
        s.zbuffer_dirty_on_host = false;
        s.zbuffer_stale_on_host = false;
    }
    return true;
}

// hand the batched triangles to the device (on an error the batch is dropped: the call that met it reports false)
inline bool submit_batch() {
    State& s = state();
    if (!s.have_batch) return true;
    double vp[16];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) vp[4 * r + c] = s.viewport_at_batch[r][c];
    const bool ok = s.ctx && TRGL_SHIM_OK(trgl_set_viewport(s.ctx, vp)) &&
                    TRGL_SHIM_OK(trgl_draw(s.ctx, s.kind, &s.uniforms, s.clip.data(), s.vary.empty() ? nullptr : s.vary.data(),
                                           s.colors.data(), s.clip.size() / 12, TRGL_MEM_HOST));
    s.clip.clear(); s.vary.clear(); s.colors.clear();
    s.have_batch = false;
    s.zbuffer_stale_on_host = true;
    return ok;
}

inline int vary_count(int kind) {
    return kind == TRGL_SHADER_GOURAUD ? TRGL_VARY_GOURAUD : (kind == TRGL_SHADER_PHONG || kind == TRGL_SHADER_EYE) ? TRGL_VARY_PHONG : 0;   // FLAT, CHECKER: 0
}
inline bool same_matrix(const mat<4, 4>& a, const mat<4, 4>& b) { return std::memcmp(&a, &b, sizeof(a)) == 0; }

}  // namespace trgl_shim

// The first C-ABI error since the last gl_clear_error() (TRGL_OK = none) and its message.
inline int gl_last_error() { return trgl_shim::state().err; }
inline const char* gl_last_error_message() { return trgl_shim::state().err_msg.c_str(); }
inline void gl_clear_error() { trgl_shim::State& s = trgl_shim::state(); s.err = TRGL_OK; s.err_msg.clear(); }

// ---- our_gl.h:25-31 ------------------------------------------------------------------------------
inline void lookat(const vec3 eye, const vec3 center, const vec3 up) {            // our_gl.cpp:25-41
    vec3 z = normalized(eye - center), x = normalized(cross(up, z)), y = cross(z, x);
    ModelView = mat<4, 4>::identity();
    for (int i = 0; i < 3; ++i) { ModelView[0][i] = x[i]; ModelView[1][i] = y[i]; ModelView[2][i] = z[i]; }
    ModelView[0][3] = -dot(x, eye); ModelView[1][3] = -dot(y, eye); ModelView[2][3] = -dot(z, eye);
}
inline void init_perspective(double fov_deg, double aspect, double znear, double zfar) {   // our_gl.cpp:44-56
    double t = std::tan(fov_deg * M_PI / 180.0 / 2.0);
    Perspective = mat<4, 4>::identity();
    Perspective[0][0] = 1.0 / (aspect * t);
    Perspective[1][1] = 1.0 / t;
    Perspective[2][2] = (zfar + znear) / (znear - zfar);
    Perspective[2][3] = (2.0 * zfar * znear) / (znear - zfar);
    Perspective[3][2] = -1.0;
    Perspective[3][3] = 0.0;
}
inline void init_viewport(int x, int y, int w, int h) {                           // our_gl.cpp:59-69
    Viewport = mat<4, 4>::identity();
    Viewport[0][0] = w / 2.0; Viewport[1][1] = h / 2.0;
    Viewport[0][3] = x + w / 2.0; Viewport[1][3] = y + h / 2.0;
    Viewport[2][2] = 1.0; Viewport[2][3] = 0.0;
}
inline void init_zbuffer(int width, int height) {                                 // our_gl.cpp:72-74
    zbuffer.assign(std::size_t(width) * height, std::numeric_limits<double>::infinity());
}

inline void trgl_zbuffer_proxy::pull() const {
    trgl_shim::State& s = trgl_shim::state();
    if (!s.ctx) return;
    trgl_shim::submit_batch();                                  // triangles batched so far are drawn against the depths as they are now
    if (!s.zbuffer_stale_on_host || s.zbuffer_dirty_on_host) return;
    std::vector<double>& hz = const_cast<std::vector<double>&>(host_);
    hz.resize(std::size_t(s.w) * s.h);
    if (TRGL_SHIM_OK(trgl_read_zbuffer(s.ctx, hz.data()))) s.zbuffer_stale_on_host = false;
}
inline void trgl_zbuffer_proxy::touched() { trgl_shim::state().zbuffer_dirty_on_host = true; }
inline void trgl_zbuffer_proxy::assign(std::size_t n, double v) {
    trgl_shim::State& s = trgl_shim::state();
    if (s.ctx) { trgl_shim::submit_batch(); (void)TRGL_SHIM_OK(trgl_flush(s.ctx)); }   // earlier draws see the old depths
    host_.assign(n, v);
    s.zbuffer_dirty_on_host = true; s.zbuffer_stale_on_host = false;
}

// Tell the shim the host changed the depths behind the proxy's back (through zbuffer.raw()) or the framebuffer's pixels.
inline void gl_zbuffer_modified() {
    trgl_shim::State& s = trgl_shim::state();
    if (s.ctx) { trgl_shim::submit_batch(); (void)TRGL_SHIM_OK(trgl_flush(s.ctx)); }   // what was batched was drawn against the OLD depths
    s.zbuffer_dirty_on_host = true;
}
inline void gl_framebuffer_modified(TGAImage& fb) {
    trgl_shim::State& s = trgl_shim::state();
    if (s.ctx) { trgl_shim::submit_batch(); (void)TRGL_SHIM_OK(trgl_write_framebuffer(s.ctx, fb.buffer())); }
}

// Model textures live on the device: TGAImage::buffer() layout, one slot per map (include/trgl.h).
inline bool gl_upload_texture(TGAImage& framebuffer, int slot, const TGAImage& img) {
    if (!trgl_shim::bind(framebuffer)) return false;
    const bool ok = trgl_shim::submit_batch();
    return TRGL_SHIM_OK(trgl_upload_texture(trgl_shim::state().ctx, slot, trgl_shim::image_bytes(img), img.width(), img.height(), trgl_shim::image_bpp(img))) && ok;
}

// ---- our_gl.h:58 ---------------------------------------------------------------------------------
inline void rasterize(const Triangle& clip, const IShader& shader, TGAImage& framebuffer) {
    using namespace trgl_shim;
    State& s = state();
    trgl_shader_desc d;
    if (!shader.describe(d)) {                                    // a programming error, not a run-time condition
        std::fprintf(stderr, "trgl: rasterize(): this IShader subclass has no device descriptor (describe()); "
                             "the fragment stage runs on the GPU and there is no CPU fallback\n");
        std::abort();
    }
    if (!bind(framebuffer)) return;                               // (reported through gl_last_error(); rasterize() is void, our_gl.h:58)
    const int K = vary_count(d.kind);
    if (s.have_batch && (s.kind != d.kind || std::memcmp(&s.uniforms, &d.uniforms, sizeof(trgl_uniforms)) != 0 ||
                         !same_matrix(s.viewport_at_batch, Viewport)))
        submit_batch();
    if (!s.have_batch) { s.have_batch = true; s.kind = d.kind; s.uniforms = d.uniforms; s.viewport_at_batch = Viewport; }
    static_assert(sizeof(Triangle) == 12 * sizeof(double), "Triangle must be 12 packed doubles");
    const double* cp = reinterpret_cast<const double*>(&clip[0]);
    s.clip.insert(s.clip.end(), cp, cp + 12);                   // the Triangle memory image (our_gl.h:55)
    if (K) s.vary.insert(s.vary.end(), d.varyings, d.varyings + K);
    s.colors.push_back(d.color);
    if (s.clip.size() >= std::size_t(12) << 20) submit_batch();      // bound host memory: 1 Mi triangles per batch
}

// A whole face loop in one call (main.cpp:660-666, 692-698, 715-721):
//     for face: for v in 0..2: clip[v] = shader.vertex(face, v);  rasterize(clip, shader, framebuffer);
// The shader's describe() supplies kind + uniforms (ModelView at call time, lights, texture slots); the vertex stage
// (main.cpp:71-90 = 199-218: eye = ModelView*(p,1), normal_eye = ModelView*(n,0), clip = Perspective*eye) runs on the
// device over the indexed mesh.  `vertices`: nv rows of `stride` doubles starting with position[3], normal[3], uv[2]
// (the reference's Vertex, model.h:14-20, has stride 14); `indices`: 3 per face.
inline bool gl_draw_indexed(const IShader& shader, const double* vertices, int stride, std::size_t nv,
                            const unsigned int* indices, std::size_t nfaces, TGAImage& framebuffer) {
    using namespace trgl_shim;
    State& s = state();
    if (!bind(framebuffer)) return false;
    bool ok = submit_batch();                                   // earlier rasterize() calls come first
    trgl_shader_desc d;
    if (!shader.describe(d) || (d.kind != TRGL_SHADER_PHONG && d.kind != TRGL_SHADER_EYE)) {
        std::fprintf(stderr, "trgl: gl_draw_indexed(): needs a PHONG or EYE shader with a device descriptor\n");
        std::abort();
    }
    double vp[16], pj[16];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { vp[4 * r + c] = Viewport[r][c]; pj[4 * r + c] = Perspective[r][c]; }
    static_assert(sizeof(unsigned int) == sizeof(std::uint32_t), "indices are 32-bit");
    ok = TRGL_SHIM_OK(trgl_set_viewport(s.ctx, vp)) &&
         TRGL_SHIM_OK(trgl_draw_indexed(s.ctx, d.kind, &d.uniforms, pj, vertices, stride, nv,
                                        reinterpret_cast<const std::uint32_t*>(indices), nfaces, TRGL_MEM_HOST)) && ok;
    s.zbuffer_stale_on_host = true;
    return ok;
}
// ... for a model that keeps `vertices` (records of packed doubles) and `indices` as the reference's Model does (model.h:114-115)
template <class ModelT> inline bool gl_draw_model(const ModelT& model, const IShader& shader, TGAImage& framebuffer) {
    using V = typename std::decay<decltype(model.vertices[0])>::type;
    static_assert(sizeof(V) % sizeof(double) == 0, "vertex records must be packed doubles");
    return gl_draw_indexed(shader, reinterpret_cast<const double*>(model.vertices.data()), int(sizeof(V) / sizeof(double)),
                    model.vertices.size(), model.indices.data(), model.indices.size() / 3, framebuffer);
}

// Run everything submitted so far and bring the pixels back into the caller's TGAImage (before framebuffer.get(),
// write_tga_file(), main.cpp:743,773).  The depths follow on demand through the `zbuffer` proxy.
// false: something submitted since the last gl_flush() failed (gl_last_error()); the pixels hold what could be drawn.
inline bool gl_flush(TGAImage& framebuffer) {
    using namespace trgl_shim;
    State& s = state();
    if (!bind(framebuffer)) return false;
    const bool ok = submit_batch();
    return TRGL_SHIM_OK(trgl_read_framebuffer(s.ctx, framebuffer.buffer())) && ok && s.err == TRGL_OK;
}

// main.cpp:751-785 on the device (z-buffer never leaves HBM): fills the three images the reference writes as
// zbuffer.tga, ao.tga and final.tga.  Any of the pointers may be null.
inline bool gl_postprocess(TGAImage& framebuffer, TGAImage* zbuffer_image, TGAImage* ao_map, TGAImage* final_result) {
    using namespace trgl_shim;
    State& s = state();
    if (!bind(framebuffer)) return false;
    const bool ok = submit_batch();
    auto prep = [&](TGAImage* img) -> std::uint8_t* {
        if (!img) return nullptr;
        if (img->width() != s.w || img->height() != s.h || image_bpp(*img) != 3) *img = TGAImage(s.w, s.h, TGAImage::RGB);
        return img->buffer();
    };
    return TRGL_SHIM_OK(trgl_postprocess(s.ctx, nullptr, prep(zbuffer_image), prep(ao_map), prep(final_result))) && ok;
}

inline void print_render_stats() {                                                // our_gl.cpp:204-210
    trgl_shim::State& s = trgl_shim::state();
    trgl_stats st{};
    if (s.ctx && (trgl_shim::submit_batch(), TRGL_SHIM_OK(trgl_get_stats(s.ctx, &st)))) { }
    else { st.min_x = st.min_y = INT32_MAX; st.max_x = st.max_y = INT32_MIN; st.min_z = std::numeric_limits<double>::infinity(); st.max_z = -st.min_z; }
    char line[1024];
    trgl_format_stats(&st, line, sizeof line);
    std::fputs(line, stderr);
}
inline void gl_shutdown() { trgl_shim::State& s = trgl_shim::state(); if (s.ctx) { trgl_destroy(s.ctx); s.ctx = nullptr; } }
