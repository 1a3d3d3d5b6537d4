// trgl_obj.h — a small Wavefront OBJ reader producing the reference's `Vertex` / index arrays (model.h:14-20,
// 114-115) for trgl_draw_indexed.  SURVEY.md §8(f) row N2: it stands in for the Assimp import of model.cpp:89-205
// (Assimp is a third-party library that is absent here).  What it reproduces of that path:
//   * triangulation of polygons (aiProcess_Triangulate) as a fan, faces in file order;
//   * FlipUVs: v -> 1 - v (model.cpp:93);
//   * float precision: Assimp holds positions / normals / uvs as `float`, model.cpp:160-175 widens them to double;
//   * one vertex per distinct (v, vt, vn) triple, in order of first use;
//   * the reference's own normal fallback generateNormalsIfNeeded() (model.cpp:269-316) when normals are missing.
// NOT reproducible without the library (parity unpinned): Assimp's number parser (last-bit differences),
// and the vertex / face reordering of JoinIdenticalVertices, ImproveCacheLocality and OptimizeMeshes (model.cpp:96-98),
// which changes triangle submission order and therefore z-tie winners and fragments_drawn.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <tuple>
#include <vector>

namespace trgl_obj {

struct Mesh {
    std::vector<double> vertices;      // nv x 14: position(3) normal(3) texcoord(2) tangent(3) bitangent(3)
    std::vector<std::uint32_t> indices;   // 3 per face
    std::string error;
};

inline bool load(const std::string& path, Mesh& out) {
    std::ifstream in(path);
    if (!in) { out.error = "cannot open " + path; return false; }
    std::vector<float> pos, nrm, tex;
    std::map<std::tuple<int, int, int>, std::uint32_t> seen;
    std::string line;
    auto resolve = [](long idx, size_t count) -> int {           // OBJ indices are 1-based; negative = relative to the end
        if (idx > 0 && size_t(idx) <= count) return int(idx - 1);
        if (idx < 0 && size_t(-(idx + 1)) < count) return int(long(count) + idx);     // -(idx + 1): idx may be LONG_MIN (found by tests/host/fuzz_parsers)
        return -1;
    };
    while (std::getline(in, line)) {
        const char* p = line.c_str();
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            char* e; p += 2;
            for (int k = 0; k < 3; ++k) { pos.push_back(float(std::strtod(p, &e))); p = e; }
        } else if (p[0] == 'v' && p[1] == 'n') {
            char* e; p += 2;
            for (int k = 0; k < 3; ++k) { nrm.push_back(float(std::strtod(p, &e))); p = e; }
        } else if (p[0] == 'v' && p[1] == 't') {
            char* e; p += 2;
            float u = float(std::strtod(p, &e)); p = e;
            float v = float(std::strtod(p, &e));
            tex.push_back(u); tex.push_back(1.0f - v);           // aiProcess_FlipUVs
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2;
            std::vector<std::uint32_t> corner;
            while (*p) {
                while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
                if (!*p) break;
                char* e;
                long vi = std::strtol(p, &e, 10), ti = 0, ni = 0;
                if (e == p) break;
                p = e;
                if (*p == '/') {
                    ++p;
                    if (*p != '/') { ti = std::strtol(p, &e, 10); p = e; }
                    if (*p == '/') { ++p; ni = std::strtol(p, &e, 10); p = e; }
                }
                int v = resolve(vi, pos.size() / 3), t = ti ? resolve(ti, tex.size() / 2) : -1, n = ni ? resolve(ni, nrm.size() / 3) : -1;
                if (v < 0) { out.error = "face references a missing vertex: " + line; return false; }
                auto key = std::make_tuple(v, t, n);
                auto it = seen.find(key);
                if (it == seen.end()) {
                    std::uint32_t id = std::uint32_t(out.vertices.size() / 14);
                    double rec[14] = { pos[3 * v], pos[3 * v + 1], pos[3 * v + 2], 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
                    if (n >= 0) { rec[3] = nrm[3 * n]; rec[4] = nrm[3 * n + 1]; rec[5] = nrm[3 * n + 2]; }
                    if (t >= 0) { rec[6] = tex[2 * t]; rec[7] = tex[2 * t + 1]; }
                    out.vertices.insert(out.vertices.end(), rec, rec + 14);
                    it = seen.emplace(key, id).first;
                }
                corner.push_back(it->second);
            }
            for (size_t k = 1; k + 1 < corner.size(); ++k) {     // fan
                out.indices.push_back(corner[0]); out.indices.push_back(corner[k]); out.indices.push_back(corner[k + 1]);
            }
        }
    }
    // generateNormalsIfNeeded(), model.cpp:269-316
    const size_t nv = out.vertices.size() / 14;
    auto nlen = [&](size_t i) { const double* n = &out.vertices[i * 14 + 3]; double s = 0; s += n[0] * n[0]; s += n[1] * n[1]; s += n[2] * n[2]; return std::sqrt(s); };
    bool needs = false;
    for (size_t i = 0; i < nv; ++i) if (nlen(i) < 0.001) { needs = true; break; }
    if (needs) {
        for (size_t i = 0; i < nv; ++i) for (int k = 0; k < 3; ++k) out.vertices[i * 14 + 3 + k] = 0;
        for (size_t f = 0; f + 2 < out.indices.size(); f += 3) {
            const double* v0 = &out.vertices[size_t(out.indices[f]) * 14];
            const double* v1 = &out.vertices[size_t(out.indices[f + 1]) * 14];
            const double* v2 = &out.vertices[size_t(out.indices[f + 2]) * 14];
            double e1[3], e2[3];
            for (int k = 0; k < 3; ++k) { e1[k] = v1[k] - v0[k]; e2[k] = v2[k] - v0[k]; }
            double fn[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
            for (int c = 0; c < 3; ++c) { double* n = &out.vertices[size_t(out.indices[f + c]) * 14 + 3]; for (int k = 0; k < 3; ++k) n[k] = n[k] + fn[k]; }
        }
        for (size_t i = 0; i < nv; ++i) {
            double* n = &out.vertices[i * 14 + 3];
            double l = nlen(i);
            if (l > 0.001) { for (int k = 0; k < 3; ++k) n[k] = n[k] / l; }
            else { n[0] = 0; n[1] = 0; n[2] = 1; }
        }
    }
    return true;
}

}  // namespace trgl_obj
