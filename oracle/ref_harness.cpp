// ref_harness.cpp — TEST INFRASTRUCTURE, NOT PRODUCT.
//
// Drives the reference's own rasterize() (compiled, unmodified, from /root/reference/our_gl.cpp +
// tgaimage.cpp where they lie — see oracle/Makefile; nothing of the reference is copied here) over
// scene files written by tests/refharness.py, and dumps framebuffer, z-buffer and the
// print_render_stats() line.  Used only in the build container to (1) validate the C restatement
// oracle/trgl_oracle.c bit-for-bit and (2) generate the golden fixtures under tests/golden/.
//
// One process per scene: the reference's diagnostic counters are file-static and never reset
// (our_gl.cpp:18-22).
//
// Shaders here are IShader subclasses (our_gl.h:36-52):
//   FLAT / GOURAUD / CHECKER are defined on the reference's own TGAColor (tgaimage.h:29-63); CHECKER is the one that discards;
//   PHONG / EYE call the C restatement's fragment (orc_fragment): main.cpp cannot be compiled here
//   (model.h includes Assimp), so those bodies are NOT pinned by this harness — what it pins for
//   them is everything around the up-call: the perspective-correct bary handed to fragment(), the
//   z-test order and TGAImage::set of the returned colour.
// Mode "vecops" evaluates geometry.h / tgaimage.h value ops on given inputs so the restatement's
// helpers can be compared with the real ones.

#include "our_gl.h"          // from -I/root/reference
#include "trgl_oracle.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <vector>

namespace {

struct Reader {
    std::vector<unsigned char> buf; size_t pos = 0;
    bool load(const char* path) {
        std::ifstream in(path, std::ios::binary);
        if (!in) return false;
        buf.assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
        return true;
    }
    template <class T> T get() { T v; std::memcpy(&v, &buf[pos], sizeof(T)); pos += sizeof(T); return v; }
    const unsigned char* take(size_t n) { const unsigned char* p = &buf[pos]; pos += n; return p; }
    void align8() { pos = (pos + 7) & ~size_t(7); }
};

TGAColor color_from_packed(uint32_t v) {
    return TGAColor((uint8_t)((v >> 16) & 0xff), (uint8_t)((v >> 8) & 0xff), (uint8_t)(v & 0xff), (uint8_t)((v >> 24) & 0xff));
}

struct FlatShader : IShader {
    TGAColor color;
    std::pair<bool, TGAColor> fragment(const vec3) const override { return { false, color }; }
};

struct GouraudShader : IShader {
    double intensity[3]; TGAColor base;
    std::pair<bool, TGAColor> fragment(const vec3 bar) const override {
        double i = intensity[0] * bar[0] + intensity[1] * bar[1] + intensity[2] * bar[2];
        return { false, base * (float)i };      // TGAColor::operator*(float), tgaimage.h:55-62
    }
};

// The discarding kind (include/trgl.h, TRGL_SHADER_CHECKER): exercises `if (discard) continue;` of the reference's rasterize()
// (our_gl.cpp:187-188) - no depth write, no colour write, no counters for a discarded fragment.
struct CheckerShader : IShader {
    TGAColor color; int cells;
    std::pair<bool, TGAColor> fragment(const vec3 bar) const override {
        const int a = (int)(bar[0] * cells), c = (int)(bar[1] * cells);
        return { ((a ^ c) & 1) != 0, color };
    }
};

struct RestatedFragShader : IShader {
    int kind; const trgl_uniforms* u; const orc_texture* tex; const double* vary;
    std::pair<bool, TGAColor> fragment(const vec3 bar) const override {
        double b[3] = { bar[0], bar[1], bar[2] };
        uint8_t bgra[4];
        int bytespp = orc_fragment(kind, u, tex, vary, 0, b, bgra);
        TGAColor c(bgra, (uint8_t)4);
        c.bytespp = (uint8_t)bytespp;
        return { false, c };
    }
};

int run_scene(const char* in_path, const char* out_path) {
    Reader r;
    if (!r.load(in_path)) { std::fprintf(stderr, "cannot read %s\n", in_path); return 2; }
    if (std::memcmp(r.take(8), "TRGSCN01", 8) != 0) { std::fprintf(stderr, "bad magic\n"); return 2; }
    int W = r.get<int32_t>(), H = r.get<int32_t>(), bpp = r.get<int32_t>();
    int ndraws = r.get<int32_t>(), ntex = r.get<int32_t>(); r.get<int32_t>();
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Viewport[i][j] = r.get<double>();
    uint8_t clear[4]; std::memcpy(clear, r.take(4), 4); r.take(4);
    double zclear = r.get<double>();

    std::vector<orc_texture> tex(TRGL_MAX_TEXTURES, orc_texture{ nullptr, 0, 0, 0 });
    for (int t = 0; t < ntex; ++t) {
        int slot = r.get<int32_t>(), w = r.get<int32_t>(), h = r.get<int32_t>(), tb = r.get<int32_t>();
        tex[slot] = orc_texture{ r.take((size_t)w * h * tb), w, h, tb };
        r.align8();
    }

    TGAColor clear_color(clear, (uint8_t)4);
    TGAImage framebuffer(W, H, bpp, clear_color);
    init_zbuffer(W, H);
    if (!(zclear == std::numeric_limits<double>::infinity()))
        for (auto& z : zbuffer) z = zclear;

    double raster_seconds = 0.0;      // time inside the per-triangle rasterize() loops only
    for (int d = 0; d < ndraws; ++d) {
        int kind = r.get<int32_t>(); r.get<int32_t>();
        uint64_t n = r.get<uint64_t>();
        trgl_uniforms u; std::memcpy(&u, r.take(sizeof(u)), sizeof(u));
        const double* clip = (const double*)r.take(n * 12 * sizeof(double));
        int K = kind == TRGL_SHADER_GOURAUD ? TRGL_VARY_GOURAUD : (kind == TRGL_SHADER_PHONG || kind == TRGL_SHADER_EYE) ? 24 : 0;
        const double* vary = (const double*)r.take(n * K * sizeof(double));
        const uint32_t* colors = (const uint32_t*)r.take(n * sizeof(uint32_t));
        r.align8();

        FlatShader flat; GouraudShader gour; RestatedFragShader rest; CheckerShader chk;
        chk.cells = u.reserved;
        rest.kind = kind; rest.u = &u; rest.tex = tex.data();
        auto t0 = std::chrono::steady_clock::now();
        for (uint64_t i = 0; i < n; ++i) {
            vec4 tri[3];
            for (int v = 0; v < 3; ++v) for (int c = 0; c < 4; ++c) tri[v][c] = clip[i * 12 + v * 4 + c];
            if (kind == TRGL_SHADER_FLAT) {
                flat.color = color_from_packed(colors[i]);
                rasterize(tri, flat, framebuffer);
            } else if (kind == TRGL_SHADER_CHECKER) {
                chk.color = color_from_packed(colors[i]);
                rasterize(tri, chk, framebuffer);
            } else if (kind == TRGL_SHADER_GOURAUD) {
                for (int v = 0; v < 3; ++v) gour.intensity[v] = vary[i * 3 + v];
                gour.base = color_from_packed(colors[i]);
                rasterize(tri, gour, framebuffer);
            } else {
                rest.vary = vary + i * 24;
                rasterize(tri, rest, framebuffer);
            }
        }
        raster_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }

    std::ostringstream captured;
    std::streambuf* old = std::cerr.rdbuf(captured.rdbuf());
    print_render_stats();                                   // our_gl.cpp:204-210
    std::cerr.rdbuf(old);
    std::string line = captured.str();

    std::ofstream out(out_path, std::ios::binary);
    size_t fb_bytes = (size_t)W * H * bpp;
    out.write((const char*)framebuffer.buffer(), fb_bytes);
    static const char pad[8] = { 0 };
    out.write(pad, (8 - fb_bytes % 8) % 8);
    out.write((const char*)zbuffer.data(), zbuffer.size() * sizeof(double));
    int32_t len = (int32_t)line.size();
    out.write((const char*)&len, 4);
    out.write(line.data(), len);
    out.write(pad, (8 - (4 + len) % 8) % 8);
    out.write((const char*)&raster_seconds, 8);
    return out ? 0 : 3;
}

// vecops: input = int32 count, then per item 3+3+16+9+3+1 doubles (v, n, M, v0v1v2, b, intensity) and
// uint32 colour; output per item: normalized(v)[3], (M*vec4(n,0)).xyz[3], v0*b0+v1*b1+v2*b2 [3],
// then 4 bytes (colour * (float)intensity).
int run_vecops(const char* in_path, const char* out_path) {
    Reader r;
    if (!r.load(in_path)) return 2;
    int count = r.get<int32_t>(); r.get<int32_t>();
    std::ofstream out(out_path, std::ios::binary);
    for (int i = 0; i < count; ++i) {
        vec3 v, n, v0, v1, v2, b; mat<4, 4> M;
        for (int k = 0; k < 3; ++k) v[k] = r.get<double>();
        for (int k = 0; k < 3; ++k) n[k] = r.get<double>();
        for (int a = 0; a < 4; ++a) for (int c = 0; c < 4; ++c) M[a][c] = r.get<double>();
        for (int k = 0; k < 3; ++k) v0[k] = r.get<double>();
        for (int k = 0; k < 3; ++k) v1[k] = r.get<double>();
        for (int k = 0; k < 3; ++k) v2[k] = r.get<double>();
        for (int k = 0; k < 3; ++k) b[k] = r.get<double>();
        double inten = r.get<double>();
        uint32_t packed = r.get<uint32_t>(); r.get<uint32_t>();
        vec3 nv = normalized(v);
        vec3 md = (M * make_vec4(n[0], n[1], n[2], 0.0)).xyz();
        vec3 ip = v0 * b[0] + v1 * b[1] + v2 * b[2];
        TGAColor sc = color_from_packed(packed) * (float)inten;
        double o[9] = { nv[0], nv[1], nv[2], md[0], md[1], md[2], ip[0], ip[1], ip[2] };
        out.write((const char*)o, sizeof(o));
        out.write((const char*)sc.bgra, 4);
        static const char pad[4] = { 0 };
        out.write(pad, 4);
    }
    return out ? 0 : 3;
}

// tga: input = int32 w,h,bpp,vflip,rle,pad + pixel bytes; output = the file TGAImage::write_tga_file produces.
int run_tga(const char* in_path, const char* out_path) {
    Reader r;
    if (!r.load(in_path)) return 2;
    int w = r.get<int32_t>(), h = r.get<int32_t>(), bpp = r.get<int32_t>(), vflip = r.get<int32_t>(), rle = r.get<int32_t>();
    r.get<int32_t>();
    TGAImage img(w, h, bpp);
    std::memcpy(img.buffer(), r.take((size_t)w * h * bpp), (size_t)w * h * bpp);
    return img.write_tga_file(out_path, vflip != 0, rle != 0) ? 0 : 3;      // tgaimage.cpp:161-191
}

// tgaread: input = a .tga file; output = int32 ok, w, h, bpp + (if ok) the TGAImage buffer after TGAImage::read_tga_file.
int run_tgaread(const char* in_path, const char* out_path) {
    TGAImage img;
    const bool ok = img.read_tga_file(in_path);                                   // tgaimage.cpp:76-126
    std::ofstream out(out_path, std::ios::binary);
    int32_t hd[4] = { ok ? 1 : 0, img.width(), img.height(), 0 };
    if (ok) hd[3] = img.get(0, 0).bytespp;                                         // TGAColor(p, bpp), tgaimage.h:47-51
    out.write((const char*)hd, sizeof hd);
    if (ok) out.write((const char*)img.buffer(), (std::streamsize)hd[1] * hd[2] * hd[3]);
    return out ? 0 : 3;
}

// sample2d: input = int32 w, h, bpp, count + texels (padded to 8) + count x 2 doubles (uv); output per sample: bgra[4], bytespp,
// 3 pad bytes, from the reference's own IShader::sample2D (our_gl.h:38-44) -> TGAImage::get (tgaimage.cpp:24-30).  Model::diffuse /
// normal / specular (model.cpp:415-459, not compilable here: Assimp) use the same clamp(int(uv * size), 0, size - 1) + get().
int run_sample2d(const char* in_path, const char* out_path) {
    Reader r;
    if (!r.load(in_path)) return 2;
    int w = r.get<int32_t>(), h = r.get<int32_t>(), bpp = r.get<int32_t>(), count = r.get<int32_t>();
    TGAImage img(w, h, bpp);
    std::memcpy(img.buffer(), r.take((size_t)w * h * bpp), (size_t)w * h * bpp);
    r.align8();
    std::ofstream out(out_path, std::ios::binary);
    for (int i = 0; i < count; ++i) {
        vec2 uv; uv.x = r.get<double>(); uv.y = r.get<double>();
        TGAColor c = IShader::sample2D(img, uv);
        unsigned char rec[8] = { c.bgra[0], c.bgra[1], c.bgra[2], c.bgra[3], c.bytespp, 0, 0, 0 };
        out.write((const char*)rec, 8);
    }
    return out ? 0 : 3;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc == 4 && std::strcmp(argv[1], "scene") == 0) return run_scene(argv[2], argv[3]);
    if (argc == 4 && std::strcmp(argv[1], "vecops") == 0) return run_vecops(argv[2], argv[3]);
    if (argc == 4 && std::strcmp(argv[1], "tga") == 0) return run_tga(argv[2], argv[3]);
    if (argc == 4 && std::strcmp(argv[1], "tgaread") == 0) return run_tgaread(argv[2], argv[3]);
    if (argc == 4 && std::strcmp(argv[1], "sample2d") == 0) return run_sample2d(argv[2], argv[3]);
    std::fprintf(stderr, "usage: ref_harness scene|vecops|tga|tgaread|sample2d <in> <out>\n");
    return 1;
}
