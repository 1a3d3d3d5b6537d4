// demo_main.cpp — a main.cpp-shaped caller (the reference's render section, main.cpp:606-792) on the MI355X
// rasterizer through the shim headers: same globals, same face loops, same rasterize() calls.
//   demo_main <model.bin> <out.bin> [out.tga [postprocess_basename]]
// model.bin is written by tests/test_shim_demo.py (a procedural head stand-in: the reference's obj/ assets are
// absent); out.bin = framebuffer bytes, z-buffer, stats line — compared with the CPU oracle by the test.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../tinyrenderder_amd/shim/trgl_shaders.h"

struct Reader {
    std::vector<unsigned char> buf; size_t pos = 0;
    bool load(const char* p) { std::ifstream in(p, std::ios::binary); if (!in) return false; buf.assign(std::istreambuf_iterator<char>(in), {}); return true; }
    template <class T> T get() { T v; std::memcpy(&v, &buf[pos], sizeof(T)); pos += sizeof(T); return v; }
    void read(void* dst, size_t n) { std::memcpy(dst, &buf[pos], n); pos += n; }
    void align8() { pos = (pos + 7) & ~size_t(7); }
};

// stands in for the reference's Model (model.h:46-131): per-face-vertex attributes + material[0] maps, and the same data as
// the indexed mesh the reference keeps (Model::vertices / indices, model.h:114-115; Vertex = model.h:14-20)
struct Vertex { vec3 position, normal; vec2 texcoord; vec3 tangent, bitangent; };
struct Model {
    std::vector<double> pos, nrm, tex;       // [nfaces][3][3], [nfaces][3][3], [nfaces][3][2]
    std::vector<Vertex> vertices;
    std::vector<unsigned int> indices;
    TGAImage diffuse, normalmap, specular;
    void build_indexed() {                   // one vertex per face corner (no sharing): the order of the faces is what matters
        vertices.resize(pos.size() / 3); indices.resize(pos.size() / 3);
        for (size_t i = 0; i < vertices.size(); ++i) {
            vertices[i].position = make_vec3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
            vertices[i].normal = make_vec3(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
            vertices[i].texcoord = make_vec2(tex[2 * i], tex[2 * i + 1]);
            indices[i] = (unsigned int)i;
        }
    }
    int nfaces() const { return int(pos.size() / 9); }
    vec3 vert(int f, int v) const { return make_vec3(pos[(f * 3 + v) * 3], pos[(f * 3 + v) * 3 + 1], pos[(f * 3 + v) * 3 + 2]); }
    vec3 normal(int f, int v) const { return make_vec3(nrm[(f * 3 + v) * 3], nrm[(f * 3 + v) * 3 + 1], nrm[(f * 3 + v) * 3 + 2]); }
    vec2 uv(int f, int v) const { return make_vec2(tex[(f * 3 + v) * 2], tex[(f * 3 + v) * 2 + 1]); }
    int diffuse_slot() const { return diffuse.width() > 0 ? 0 : -1; }
    int normal_slot() const { return normalmap.width() > 0 ? 1 : -1; }
    int specular_slot() const { return specular.width() > 0 ? 2 : -1; }
};
using PhongShader = PhongShaderT<Model>;
using EyeShader = EyeShaderT<Model>;

static TGAImage read_texture(Reader& r) {
    int w = r.get<int32_t>(), h = r.get<int32_t>(), bpp = r.get<int32_t>(); r.get<int32_t>();
    if (w <= 0) return TGAImage();
    TGAImage img(w, h, bpp);
    r.read(img.buffer(), size_t(w) * h * bpp); r.align8();
    return img;
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: demo_main <model.bin> <out.bin> [out.tga [postprocess_basename [loop]]]\n"); return 1; }
    const bool use_draw_model = !(argc > 5 && std::strcmp(argv[5], "loop") == 0);    // "loop": the reference's per-face rasterize() loop
    Reader r;
    if (!r.load(argv[1]) || std::memcmp(&r.buf[0], "TRGMDL01", 8) != 0) { std::fprintf(stderr, "bad model file\n"); return 2; }
    r.pos = 8;
    const int WIDTH = r.get<int32_t>(), HEIGHT = r.get<int32_t>(), bpp = r.get<int32_t>(), nfaces = r.get<int32_t>();
    mat<4, 4> view, proj;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) view[i][j] = r.get<double>();
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) proj[i][j] = r.get<double>();
    vec3 key_light_dir, fill_light_dir, rim_light_dir;
    for (int i = 0; i < 3; ++i) key_light_dir[i] = r.get<double>();
    for (int i = 0; i < 3; ++i) fill_light_dir[i] = r.get<double>();
    for (int i = 0; i < 3; ++i) rim_light_dir[i] = r.get<double>();
    const double strength = r.get<double>();
    Model head;
    head.pos.resize(size_t(nfaces) * 9); head.nrm.resize(size_t(nfaces) * 9); head.tex.resize(size_t(nfaces) * 6);
    r.read(head.pos.data(), head.pos.size() * 8); r.read(head.nrm.data(), head.nrm.size() * 8); r.read(head.tex.data(), head.tex.size() * 8);
    head.diffuse = read_texture(r); head.normalmap = read_texture(r); head.specular = read_texture(r);
    head.build_indexed();
    const int nflat = r.get<int32_t>(); r.get<int32_t>();
    std::vector<double> flat_clip(size_t(nflat) * 12); std::vector<uint32_t> flat_col(nflat);
    r.read(flat_clip.data(), flat_clip.size() * 8); r.read(flat_col.data(), flat_col.size() * 4);

    // ---- main.cpp:606-612 ----
    TGAImage framebuffer(WIDTH, HEIGHT, bpp);
    init_zbuffer(WIDTH, HEIGHT);
    ModelView = view;
    Perspective = proj;
    init_viewport(0, 0, WIDTH, HEIGHT);
    gl_upload_texture(framebuffer, 0, head.diffuse);
    gl_upload_texture(framebuffer, 1, head.normalmap);
    gl_upload_texture(framebuffer, 2, head.specular);

    // ---- head pass, main.cpp:685-698 ----
    PhongShader head_shader(&head);
    head_shader.initLightDirections(key_light_dir, fill_light_dir, rim_light_dir);
    head_shader.normal_map_strength = strength;
    if (use_draw_model) {
        gl_draw_model(head, head_shader, framebuffer);       // the whole face loop in one call, vertex stage on the device
    } else {
        for (int face = 0; face < head.nfaces(); ++face) {
            vec4 clip_space_triangle[3];
            for (int vertex = 0; vertex < 3; ++vertex) clip_space_triangle[vertex] = head_shader.vertex(face, vertex);
            rasterize(clip_space_triangle, head_shader, framebuffer);
        }
    }
    std::vector<double> zbuffer_before_eyes = zbuffer;       // main.cpp:700, unchanged: the proxy completes the pending draws

    // ---- eyes pass, main.cpp:711-721 (every third face of the same mesh stands in for the eye model) ----
    EyeShader eye_shader(&head);
    eye_shader.initLightDirections(key_light_dir, rim_light_dir);
    for (int face = 0; face < head.nfaces(); face += 3) {
        vec4 clip_space_triangle[3];
        for (int vertex = 0; vertex < 3; ++vertex) clip_space_triangle[vertex] = eye_shader.vertex(face, vertex);
        rasterize(clip_space_triangle, eye_shader, framebuffer);
    }
    zbuffer = zbuffer_before_eyes;                           // main.cpp:730, unchanged: the eyes are drawn before the depths are replaced

    // ---- an overlay of flat triangles (BASELINE config 0's "flat shader") ----
    FlatShader flat;
    for (int i = 0; i < nflat; ++i) {
        vec4 tri[3];
        for (int v = 0; v < 3; ++v) for (int c = 0; c < 4; ++c) tri[v][c] = flat_clip[size_t(i) * 12 + v * 4 + c];
        const uint32_t p = flat_col[i];
        flat.color = TGAColor(uint8_t(p >> 16), uint8_t(p >> 8), uint8_t(p), uint8_t(p >> 24));
        rasterize(tri, flat, framebuffer);
    }
    if (!gl_flush(framebuffer)) {                            // a C-ABI error anywhere above ends up here, not in abort()
        std::fprintf(stderr, "demo_main: %s (code %d)\n", gl_last_error_message(), gl_last_error());
        return 4;
    }
    if (argc > 3) framebuffer.write_tga_file(argv[3]);       // main.cpp:743
    if (argc > 4) {                                          // main.cpp:751-785: zbuffer.tga, ao.tga, final.tga
        TGAImage zimg, ao_map, final_result;
        gl_postprocess(framebuffer, &zimg, &ao_map, &final_result);
        const std::string base = argv[4];
        zimg.write_tga_file(base + "_zbuffer.tga");
        ao_map.write_tga_file(base + "_ao.tga");
        final_result.write_tga_file(base + "_final.tga");
    }

    // stats line exactly as print_render_stats() prints it (our_gl.cpp:204-210), captured for the test
    trgl_stats st{};
    trgl_get_stats(trgl_shim::state().ctx, &st);
    char line[256];
    trgl_format_stats(&st, line, sizeof line);
    print_render_stats();

    std::ofstream out(argv[2], std::ios::binary);
    out.write(reinterpret_cast<const char*>(framebuffer.buffer()), std::streamsize(size_t(WIDTH) * HEIGHT * bpp));
    const std::vector<double>& depths = zbuffer;
    out.write(reinterpret_cast<const char*>(depths.data()), std::streamsize(depths.size() * 8));
    out.write(line, std::streamsize(std::strlen(line)));
    gl_shutdown();
    return out ? 0 : 3;
}
