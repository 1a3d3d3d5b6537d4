#define TRGL_GEOMETRY_HEADER "geometry.h"
#define TRGL_IMAGE_HEADER "tgaimage.h"
#include "trgl_shaders.h"
struct M { std::vector<vec3> vertices; std::vector<unsigned int> indices; vec3 vert(int,int) const { return vec3(); } vec3 normal(int,int) const { return vec3(); } vec2 uv(int,int) const { return vec2(); }
           int diffuse_slot() const { return 0; } int normal_slot() const { return 1; } int specular_slot() const { return 2; } };
int main() {
    TGAImage framebuffer(64, 64, TGAImage::RGB);
    init_zbuffer(64, 64); init_viewport(0, 0, 64, 64);
    M m; PhongShaderT<M> sh(&m);
    vec4 clip[3];
    for (int v = 0; v < 3; ++v) clip[v] = sh.vertex(0, v);
    rasterize(clip, sh, framebuffer);
    std::vector<double> saved = zbuffer; zbuffer = saved;
    double d = zbuffer[5]; zbuffer[6] = d + 1.0; zbuffer[7] = zbuffer[6];       // main.cpp:759-style element reads, and writes
    CheckerShader chk; chk.color = TGAColor(1, 2, 3); rasterize(clip, chk, framebuffer);
    gl_draw_model(m, sh, framebuffer);
    if (!gl_flush(framebuffer)) return gl_last_error();
    print_render_stats();
    return 0;
}
