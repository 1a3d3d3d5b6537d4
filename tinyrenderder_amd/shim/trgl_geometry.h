// trgl_geometry.h — the value types a main.cpp-shaped caller needs (vec2/vec3/vec4, mat<R,C>),
// with the reference's names and evaluation order (geometry.h:13-246) so host-side vertex stages
// give bit-identical clip coordinates and varyings.  Written from scratch; only the interface
// (names, member access, left-to-right summation from 0) follows the reference.
#pragma once
#include <cassert>
#include <cmath>
#include <ostream>

template <int n> struct vec {
    double data[n] = {};
    double& operator[](int i) { assert(i >= 0 && i < n); return data[i]; }
    double operator[](int i) const { assert(i >= 0 && i < n); return data[i]; }
};
template <> struct vec<2> {
    double x = 0, y = 0;
    double& operator[](int i) { assert(i >= 0 && i < 2); return i ? y : x; }
    double operator[](int i) const { assert(i >= 0 && i < 2); return i ? y : x; }
};
template <> struct vec<3> {
    double x = 0, y = 0, z = 0;
    double& operator[](int i) { assert(i >= 0 && i < 3); return i == 0 ? x : (i == 1 ? y : z); }
    double operator[](int i) const { assert(i >= 0 && i < 3); return i == 0 ? x : (i == 1 ? y : z); }
};
template <> struct vec<4> {
    double data[4] = { 0, 0, 0, 0 };
    double& operator[](int i) { assert(i >= 0 && i < 4); return data[i]; }
    double operator[](int i) const { assert(i >= 0 && i < 4); return data[i]; }
    double x() const { return data[0]; }
    double y() const { return data[1]; }
    double z() const { return data[2]; }
    double w() const { return data[3]; }
    vec<2> xy() const { vec<2> r; r.x = data[0]; r.y = data[1]; return r; }
    vec<3> xyz() const { vec<3> r; r.x = data[0]; r.y = data[1]; r.z = data[2]; return r; }
};
typedef vec<2> vec2;
typedef vec<3> vec3;
typedef vec<4> vec4;

template <int n> vec<n> operator+(const vec<n>& a, const vec<n>& b) { vec<n> r; for (int i = 0; i < n; ++i) r[i] = a[i] + b[i]; return r; }
template <int n> vec<n> operator-(const vec<n>& a, const vec<n>& b) { vec<n> r; for (int i = 0; i < n; ++i) r[i] = a[i] - b[i]; return r; }
template <int n> vec<n> operator*(const vec<n>& a, double s) { vec<n> r; for (int i = 0; i < n; ++i) r[i] = a[i] * s; return r; }
template <int n> vec<n> operator*(double s, const vec<n>& a) { return a * s; }
template <int n> vec<n> operator/(const vec<n>& a, double s) { vec<n> r; for (int i = 0; i < n; ++i) r[i] = a[i] / s; return r; }
template <int n> vec<n> operator-(const vec<n>& a) { return a * -1.0; }
template <int n> double dot(const vec<n>& a, const vec<n>& b) { double s = 0; for (int i = 0; i < n; ++i) s += a[i] * b[i]; return s; }
template <int n> double norm(const vec<n>& a) { return std::sqrt(dot(a, a)); }
inline vec3 normalized(const vec3& v) { double l = norm<3>(v); return l == 0 ? v : v / l; }
inline vec3 cross(const vec3& a, const vec3& b) {
    vec3 r; r.x = a[1] * b[2] - a[2] * b[1]; r.y = a[2] * b[0] - a[0] * b[2]; r.z = a[0] * b[1] - a[1] * b[0]; return r;
}
inline vec2 make_vec2(double x, double y) { vec2 r; r.x = x; r.y = y; return r; }
inline vec3 make_vec3(double x, double y, double z) { vec3 r; r.x = x; r.y = y; r.z = z; return r; }
inline vec4 make_vec4(double x, double y, double z, double w) { vec4 r; r[0] = x; r[1] = y; r[2] = z; r[3] = w; return r; }

template <int R, int C> struct mat {
    vec<C> rows[R];
    vec<C>& operator[](int r) { assert(r >= 0 && r < R); return rows[r]; }
    const vec<C>& operator[](int r) const { assert(r >= 0 && r < R); return rows[r]; }
    static mat identity() { mat m; for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) m[r][c] = (r == c) ? 1.0 : 0.0; return m; }
    mat<C, R> transpose() const { mat<C, R> t; for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) t[c][r] = rows[r][c]; return t; }
};
template <int R, int C> vec<R> operator*(const mat<R, C>& m, const vec<C>& v) { vec<R> r; for (int i = 0; i < R; ++i) r[i] = dot<C>(m[i], v); return r; }
template <int R1, int C1, int C2> mat<R1, C2> operator*(const mat<R1, C1>& a, const mat<C1, C2>& b) {
    mat<R1, C2> r;
    for (int i = 0; i < R1; ++i) for (int j = 0; j < C2; ++j) { r[i][j] = 0; for (int k = 0; k < C1; ++k) r[i][j] += a[i][k] * b[k][j]; }
    return r;
}
template <int n> std::ostream& operator<<(std::ostream& o, const vec<n>& v) { for (int i = 0; i < n; ++i) o << v[i] << " "; return o; }
