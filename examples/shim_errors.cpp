// shim_errors.cpp - what a C++ caller of the shim sees when the C ABI refuses something (tests/test_shim_demo.py).
// A flush of 70 000 full-screen triangles at 8192 x 8192 would hold 4.6e9 triangle-tile pairs: the C ABI returns
// TRGL_E_UNSUPPORTED for it (include/trgl.h), the shim drops the batch, gl_flush() returns false and gl_last_error() says why -
// and the process goes on: the next, ordinary triangle is drawn as if nothing had happened.
#include <cstdio>
#include "../tinyrenderder_amd/shim/trgl_shaders.h"

int main() {
    const int W = 8192, H = 8192;
    TGAImage framebuffer(W, H, TGAImage::RGB);
    init_zbuffer(W, H);
    init_viewport(0, 0, W, H);
    FlatShader flat; flat.color = TGAColor(10, 20, 30, 255);
    vec4 tri[3];
    const double v[3][2] = { { -1.0, -1.0 }, { 1.0, -1.0 }, { 0.0, 1.0 } };
    for (int k = 0; k < 3; ++k) { tri[k][0] = v[k][0]; tri[k][1] = v[k][1]; tri[k][2] = 0.5; tri[k][3] = 1.0; }
    for (int i = 0; i < 70000; ++i) rasterize(tri, flat, framebuffer);
    const bool ok1 = gl_flush(framebuffer);
    std::printf("flush 1: %s, code %d, message: %s\n", ok1 ? "ok" : "failed", gl_last_error(), gl_last_error_message());
    if (ok1 || gl_last_error() != TRGL_E_UNSUPPORTED) return 1;
    gl_clear_error();
    // the context is still usable
    flat.color = TGAColor(200, 100, 50, 255);
    rasterize(tri, flat, framebuffer);
    const bool ok2 = gl_flush(framebuffer);
    const TGAColor c = framebuffer.get(W / 2, H / 2);
    std::printf("flush 2: %s, centre pixel %d %d %d\n", ok2 ? "ok" : "failed", c.bgra[2], c.bgra[1], c.bgra[0]);
    gl_shutdown();
    return (ok2 && c.bgra[2] == 200 && c.bgra[1] == 100 && c.bgra[0] == 50) ? 0 : 2;
}
