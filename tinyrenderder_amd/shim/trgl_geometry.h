// trgl_geometry.h — the small value types the shim and the demo caller use: vec2 / vec3 / vec4 and mat<R,C>.
//
// A maintainer who puts the library under the reference's main.cpp keeps the reference's own geometry.h: define
// TRGL_GEOMETRY_HEADER to its path before including trgl_gl.h and this file is never seen.  It exists so that the
// demo (examples/demo_main.cpp) and the tests build without the reference tree.  Only what the rasterize() path needs
// is here, and what matters for bit-identical clip coordinates and varyings is kept: dot products are summed left to
// right starting from 0.0, normalized() divides by the length, mat * vec is one dot product per row.
//
// Layout: every vec<N> is N packed doubles (vec<4>[3] is the 12-double `Triangle` memory image the C ABI takes), every
// mat<R,C> is R*C packed doubles in row-major order.
#pragma once
#include <cmath>
#include <cstddef>

namespace trgl_math {
// storage with named members for 2 and 3 components (callers write uv.x, n.z), plain arrays otherwise
template <int N> struct store { double e[N]; };
template <> struct store<2> { union { double e[2]; struct { double x, y; }; }; };
template <> struct store<3> { union { double e[3]; struct { double x, y, z; }; }; };
}  // namespace trgl_math

template <int N> struct vec : trgl_math::store<N> {
    vec() { for (int i = 0; i < N; ++i) this->e[i] = 0.0; }
    double& operator[](int i) { return this->e[i]; }
    const double& operator[](int i) const { return this->e[i]; }
    // leading components as a shorter vector: v.xyz() of a vec4, v.xy() of a vec3 / vec4
    vec<3> xyz() const { static_assert(N >= 3, "xyz() needs three components"); vec<3> r; r.e[0] = this->e[0]; r.e[1] = this->e[1]; r.e[2] = this->e[2]; return r; }
    vec<2> xy() const { vec<2> r; r.e[0] = this->e[0]; r.e[1] = this->e[1]; return r; }
};
using vec2 = vec<2>;
using vec3 = vec<3>;
using vec4 = vec<4>;

inline vec2 make_vec2(double a, double b) { vec2 r; r[0] = a; r[1] = b; return r; }
inline vec3 make_vec3(double a, double b, double c) { vec3 r; r[0] = a; r[1] = b; r[2] = c; return r; }
inline vec4 make_vec4(double a, double b, double c, double d) { vec4 r; r[0] = a; r[1] = b; r[2] = c; r[3] = d; return r; }

// component-wise arithmetic through one helper
namespace trgl_math {
template <int N, class F> inline vec<N> zip(const vec<N>& a, const vec<N>& b, F f) { vec<N> r; for (int i = 0; i < N; ++i) r[i] = f(a[i], b[i]); return r; }
template <int N, class F> inline vec<N> map(const vec<N>& a, F f) { vec<N> r; for (int i = 0; i < N; ++i) r[i] = f(a[i]); return r; }
}  // namespace trgl_math
template <int N> inline vec<N> operator+(const vec<N>& a, const vec<N>& b) { return trgl_math::zip(a, b, [](double p, double q) { return p + q; }); }
template <int N> inline vec<N> operator-(const vec<N>& a, const vec<N>& b) { return trgl_math::zip(a, b, [](double p, double q) { return p - q; }); }
template <int N> inline vec<N> operator*(const vec<N>& a, double s) { return trgl_math::map(a, [s](double p) { return p * s; }); }
template <int N> inline vec<N> operator*(double s, const vec<N>& a) { return a * s; }
template <int N> inline vec<N> operator/(const vec<N>& a, double s) { return trgl_math::map(a, [s](double p) { return p / s; }); }
template <int N> inline vec<N> operator-(const vec<N>& a) { return a * -1.0; }

// sum of products, left to right, starting from 0.0 (the accumulation order decides the low bits)
template <int N> inline double dot(const vec<N>& a, const vec<N>& b) {
    double acc = 0.0;
    for (int i = 0; i < N; ++i) acc += a[i] * b[i];
    return acc;
}
template <int N> inline double norm(const vec<N>& a) { return std::sqrt(dot(a, a)); }
inline vec3 normalized(const vec3& a) {
    const double len = norm(a);
    if (len == 0.0) return a;
    return a / len;
}
inline vec3 cross(const vec3& a, const vec3& b) {
    return make_vec3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}

// R x C doubles, row-major; m[r] is the r-th row as a vec<C>
template <int R, int C> struct mat {
    vec<C> row[R];
    vec<C>& operator[](int r) { return row[r]; }
    const vec<C>& operator[](int r) const { return row[r]; }
    static mat identity() {
        mat m;
        for (int i = 0; i < (R < C ? R : C); ++i) m.row[i][i] = 1.0;
        return m;
    }
};
template <int R, int C> inline vec<R> operator*(const mat<R, C>& m, const vec<C>& v) {
    vec<R> out;
    for (int r = 0; r < R; ++r) out[r] = dot(m[r], v);
    return out;
}
template <int R, int K, int C> inline mat<R, C> operator*(const mat<R, K>& a, const mat<K, C>& b) {
    mat<R, C> out;
    for (int r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c) {
            double acc = 0.0;
            for (int k = 0; k < K; ++k) acc += a[r][k] * b[k][c];
            out[r][c] = acc;
        }
    return out;
}
static_assert(sizeof(vec<4>) == 4 * sizeof(double) && sizeof(vec<3>) == 3 * sizeof(double) && sizeof(vec<2>) == 2 * sizeof(double),
              "vec<N> must be N packed doubles");
