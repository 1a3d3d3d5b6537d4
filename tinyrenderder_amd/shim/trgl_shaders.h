// trgl_shaders.h — the shader kinds the device implements, as IShader subclasses with the reference's member
// names (main.cpp:39-90, 176-218) so a main.cpp-shaped face loop compiles unchanged:
//     for v in 0..2: clip[v] = shader.vertex(face, v);  rasterize(clip, shader, framebuffer);
// vertex() runs on the host (3 mat*vec per vertex, as in the reference); fragment() runs on the GPU, selected by
// describe().  `ModelT` is anything with vert(face,v), normal(face,v), uv(face,v) and texture slots.
#pragma once
#include "trgl_gl.h"

struct FlatShader : IShader {
    TGAColor color;
    bool describe(trgl_shader_desc& d) const override { d.kind = TRGL_SHADER_FLAT; d.color = trgl_shim::pack_bgra(color); return true; }
};

// the shader that DISCARDS (our_gl.h:51: fragment() returns { true, ... } and rasterize() skips the fragment, our_gl.cpp:187-188):
// a flat colour on the even cells of a cells x cells checker over the perspective-correct barycentrics (include/trgl.h)
struct CheckerShader : IShader {
    TGAColor color; int cells = 8;
    bool describe(trgl_shader_desc& d) const override {
        d.kind = TRGL_SHADER_CHECKER; d.color = trgl_shim::pack_bgra(color); d.uniforms.reserved = cells; return true;
    }
};

// classic tinyrenderer Gouraud: per-vertex intensity, colour = base * intensity (TGAColor::operator*, tgaimage.h:55-62)
struct GouraudShader : IShader {
    TGAColor base = TGAColor(255, 255, 255);
    double varying_intensity[3] = { 0, 0, 0 };
    bool describe(trgl_shader_desc& d) const override {
        d.kind = TRGL_SHADER_GOURAUD; d.color = trgl_shim::pack_bgra(base); d.varyings = varying_intensity; return true;
    }
};

namespace trgl_shim {
inline void fill_lights(trgl_uniforms& u, const vec3& key, const vec3& fill, const vec3& rim) {
    for (int i = 0; i < 3; ++i) { u.key_light_dir_eye[i] = key[i]; u.fill_light_dir_eye[i] = fill[i]; u.rim_light_dir_eye[i] = rim[i]; }
}
inline vec3 light_to_eye(const vec3& world) {                      // main.cpp:58-68: upper-left 3x3 of ModelView
    mat<3, 3> nm;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) nm[i][j] = ModelView[i][j];
    return normalized(nm * world);
}
}  // namespace trgl_shim

template <class ModelT> struct PhongShaderT : IShader {            // main.cpp:39-171
    const ModelT* model;
    vec3 key_light_dir_eye, fill_light_dir_eye, rim_light_dir_eye;
    // one contiguous block in the order include/trgl.h documents: uv[3], position_eye[3], normal_eye[3]
    struct { vec2 varying_uv[3]; vec3 varying_position_eye[3]; vec3 varying_normal_eye[3]; } v;
    double normal_map_strength = 1.0;
    explicit PhongShaderT(const ModelT* m) : model(m) {}
    void initLightDirections(const vec3& key, const vec3& fill, const vec3& rim) {
        key_light_dir_eye = trgl_shim::light_to_eye(key);
        fill_light_dir_eye = trgl_shim::light_to_eye(fill);
        rim_light_dir_eye = trgl_shim::light_to_eye(rim);
    }
    vec4 vertex(int face, int nth) override {                      // main.cpp:71-90
        vec3 p = model->vert(face, nth), n = model->normal(face, nth);
        v.varying_uv[nth] = model->uv(face, nth);
        vec4 pe = ModelView * make_vec4(p[0], p[1], p[2], 1.0);
        v.varying_position_eye[nth] = pe.xyz();
        v.varying_normal_eye[nth] = (ModelView * make_vec4(n[0], n[1], n[2], 0.0)).xyz();
        return Perspective * pe;
    }
    bool describe(trgl_shader_desc& d) const override {
        static_assert(sizeof(v) == 24 * sizeof(double), "varyings block must be 24 packed doubles");
        d.kind = TRGL_SHADER_PHONG;
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) d.uniforms.model_view[4 * r + c] = ModelView[r][c];  // main.cpp:116
        trgl_shim::fill_lights(d.uniforms, key_light_dir_eye, fill_light_dir_eye, rim_light_dir_eye);
        d.uniforms.normal_map_strength = normal_map_strength;
        d.uniforms.tex_diffuse = model->diffuse_slot(); d.uniforms.tex_normal = model->normal_slot();
        d.uniforms.tex_specular = model->specular_slot(); d.uniforms.reserved = 0;
        d.varyings = reinterpret_cast<const double*>(&v);
        return true;
    }
};

template <class ModelT> struct EyeShaderT : IShader {              // main.cpp:176-262
    const ModelT* model;
    vec3 key_light_dir_eye, rim_light_dir_eye;
    struct { vec2 varying_uv[3]; vec3 varying_position_eye[3]; vec3 varying_normal_eye[3]; } v;
    explicit EyeShaderT(const ModelT* m) : model(m) {}
    void initLightDirections(const vec3& key, const vec3& rim) {
        key_light_dir_eye = trgl_shim::light_to_eye(key);
        rim_light_dir_eye = trgl_shim::light_to_eye(rim);
    }
    vec4 vertex(int face, int nth) override {                      // main.cpp:199-218
        vec3 p = model->vert(face, nth), n = model->normal(face, nth);
        v.varying_uv[nth] = model->uv(face, nth);
        vec4 pe = ModelView * make_vec4(p[0], p[1], p[2], 1.0);
        v.varying_position_eye[nth] = pe.xyz();
        v.varying_normal_eye[nth] = (ModelView * make_vec4(n[0], n[1], n[2], 0.0)).xyz();
        return Perspective * pe;
    }
    bool describe(trgl_shader_desc& d) const override {
        d.kind = TRGL_SHADER_EYE;
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) d.uniforms.model_view[4 * r + c] = ModelView[r][c];
        trgl_shim::fill_lights(d.uniforms, key_light_dir_eye, vec3(), rim_light_dir_eye);
        d.uniforms.normal_map_strength = 1.0;
        d.uniforms.tex_diffuse = model->diffuse_slot(); d.uniforms.tex_normal = -1;
        d.uniforms.tex_specular = model->specular_slot(); d.uniforms.reserved = 0;
        d.varyings = reinterpret_cast<const double*>(&v);
        return true;
    }
};
