"""Seeded synthetic workloads for the rasterize() hot path.

The reference ships no assets (main.cpp:28-30 names obj/ files that are absent), so every
configuration in BASELINE.json is driven by procedural inputs.  Everything here is built from
SplitMix64 plus IEEE add/sub/mul/div/sqrt only (no sin/cos/pow, no BLAS), so the generated arrays
are bit-identical on every machine: golden fixtures can store a seed and a hash instead of the data.

Outputs are the memory images the C ABI takes (include/trgl.h):
  clip     float64 [n,12]  the reference's `Triangle` = vec<4>[3]  (our_gl.h:55)
  varyings float64 [n,K]   snapshot of the shader's varying_* members (main.cpp:47-49)
  colors   uint32  [n]     b | g<<8 | r<<16 | a<<24
"""
from __future__ import annotations

import hashlib

import numpy as np

SEED_DEFAULT = 0x5EED0001
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


class SplitMix64:
    """Counter-based SplitMix64: draw i of a stream is mix(seed + (i+1)*golden)."""

    def __init__(self, seed: int):
        self.seed = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        self.count = 0

    def u64(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            idx = np.arange(self.count + 1, self.count + n + 1, dtype=np.uint64)
            z = self.seed + idx * _GOLDEN
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            z = z ^ (z >> np.uint64(31))
        self.count += n
        return z

    def uniform(self, n: int, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
        u = (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return lo + (hi - lo) * u


def digest(a: np.ndarray) -> str:
    """sha256 of the raw bytes (fixtures store this instead of MB-sized buffers)."""
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).tobytes()).hexdigest()


def init_viewport(x: int, y: int, w: int, h: int) -> np.ndarray:
    """init_viewport (our_gl.cpp:59-69) as a row-major 4x4."""
    m = np.eye(4, dtype=np.float64)
    m[0, 0] = w / 2.0
    m[1, 1] = h / 2.0
    m[0, 3] = x + w / 2.0
    m[1, 3] = y + h / 2.0
    m[2, 2] = 1.0
    m[2, 3] = 0.0
    return m


def pack_bgra(b, g, r, a=255) -> np.ndarray:
    b = np.asarray(b, dtype=np.uint32)
    return (b | (np.asarray(g, np.uint32) << 8) | (np.asarray(r, np.uint32) << 16) | (np.asarray(a, np.uint32) << 24)).astype(np.uint32)


# --------------------------------------------------------------------------------------------
# C4-style random triangles: centre uniform in NDC, circumradius r px in [rmin,
# rmax) spread evenly over octaves, CCW winding (cross > 0 with +y up, our_gl.cpp:124-127),
# per-vertex z uniform in [-1,1), w = 1 (or uniform [0.5,2] for the perspective variant).
# --------------------------------------------------------------------------------------------
_C120, _S120 = -0.5, 0.8660254037844386


def random_triangles(n: int, width: int, height: int, seed: int = SEED_DEFAULT, rmin: float = 1.0,
                     rmax: float = 16.0, perspective_w: bool = False, chunk: int = 1 << 20):
    """Returns (clip [n,12] f64, colors [n] u32).  Colour = low 24 bits of the triangle id."""
    clip = np.empty((n, 12), dtype=np.float64)
    octaves = max(1, int(round(np.log2(rmax / rmin))))
    done = 0
    rng = SplitMix64(seed)
    while done < n:
        m = min(chunk, n - done)
        u = rng.uniform(m * 16).reshape(16, m)
        cx = u[0] * 2.0 - 1.0
        cy = u[1] * 2.0 - 1.0
        k = np.floor(u[2] * octaves)
        r_px = rmin * np.ldexp(1.0 + u[3], k.astype(np.int64))  # [rmin, rmax), even per octave
        rx = r_px * (2.0 / width)      # radius in NDC units
        ry = r_px * (2.0 / height)
        a = u[4] * 2.0 - 1.0
        b = u[5] * 2.0 - 1.0
        nrm = np.sqrt(a * a + b * b)
        bad = nrm == 0.0
        a = np.where(bad, 1.0, a); b = np.where(bad, 0.0, b); nrm = np.where(bad, 1.0, nrm)
        d0x, d0y = a / nrm, b / nrm
        d1x, d1y = d0x * _C120 - d0y * _S120, d0x * _S120 + d0y * _C120
        d2x, d2y = d1x * _C120 - d1y * _S120, d1x * _S120 + d1y * _C120
        out = clip[done:done + m]
        for v, (dx, dy) in enumerate(((d0x, d0y), (d1x, d1y), (d2x, d2y))):
            s = 0.7 + 0.6 * u[6 + 2 * v]           # radial jitter
            t = 0.6 * u[7 + 2 * v] - 0.3           # tangential jitter
            vx = cx + rx * (s * dx - t * dy)
            vy = cy + ry * (s * dy + t * dx)
            vz = u[12 + v] * 2.0 - 1.0
            if perspective_w:
                w = 0.5 + 1.5 * u[15] if v == 0 else 0.5 + 1.5 * ((u[15] * (v + 2.0)) % 1.0)
            else:
                w = np.ones(m)
            out[:, 4 * v + 0] = vx * w
            out[:, 4 * v + 1] = vy * w
            out[:, 4 * v + 2] = vz * w
            out[:, 4 * v + 3] = w
        done += m
    colors = (np.arange(n, dtype=np.uint64) & np.uint64(0xFFFFFF)).astype(np.uint32) | np.uint32(0xFF000000)
    return clip, colors


# --------------------------------------------------------------------------------------------
# Stand-in mesh for african_head.obj: a displaced icosphere with analytic normals and octahedral
# UVs, pushed through a look-at + OpenGL-style perspective on the host exactly as
# PhongShader::vertex does (main.cpp:71-90): eye = MV*(p,1), normal_eye = MV*(n,0), clip = P*eye.
# --------------------------------------------------------------------------------------------
def _normalize_rows(v: np.ndarray) -> np.ndarray:
    length = np.sqrt((v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]) + v[..., 2] * v[..., 2])
    return v / length[..., None]


def icosphere(level: int) -> np.ndarray:
    """[F,3,3] unit-sphere triangles (CCW seen from outside), F = 20*4**level."""
    phi = (1.0 + np.sqrt(5.0)) / 2.0
    v = np.array([[-1, phi, 0], [1, phi, 0], [-1, -phi, 0], [1, -phi, 0], [0, -1, phi], [0, 1, phi],
                  [0, -1, -phi], [0, 1, -phi], [phi, 0, -1], [phi, 0, 1], [-phi, 0, -1], [-phi, 0, 1]], dtype=np.float64)
    v = _normalize_rows(v)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]])
    tris = v[f]
    for _ in range(level):
        a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
        ab = _normalize_rows((a + b) * 0.5)
        bc = _normalize_rows((b + c) * 0.5)
        ca = _normalize_rows((c + a) * 0.5)
        tris = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1),
                               np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)], 0)
    return tris


def _oct_uv(n: np.ndarray) -> np.ndarray:
    s = np.abs(n[..., 0]) + np.abs(n[..., 1]) + np.abs(n[..., 2])
    px, py = n[..., 0] / s, n[..., 1] / s
    neg = n[..., 2] < 0
    fx = np.where(neg, (1.0 - np.abs(py)) * np.where(px >= 0, 1.0, -1.0), px)
    fy = np.where(neg, (1.0 - np.abs(px)) * np.where(py >= 0, 1.0, -1.0), py)
    return np.stack([fx * 0.5 + 0.5, fy * 0.5 + 0.5], -1)


def _matvec(m: np.ndarray, x, y, z, w):
    """mat<4,4> * vec4 with the reference's summation order (geometry.h:122-127,186-192)."""
    return [(((0.0 + m[r, 0] * x) + m[r, 1] * y) + m[r, 2] * z) + m[r, 3] * w for r in range(4)]


def lookat(eye, center, up) -> np.ndarray:
    """lookat (our_gl.cpp:25-41) as a row-major 4x4."""
    eye, center, up = (np.asarray(t, np.float64) for t in (eye, center, up))
    z = _normalize_rows(eye - center)
    x = _normalize_rows(np.cross(up, z))
    y = np.cross(z, x)
    m = np.eye(4)
    m[0, :3], m[1, :3], m[2, :3] = x, y, z
    m[0, 3] = -((x[0] * eye[0] + x[1] * eye[1]) + x[2] * eye[2])
    m[1, 3] = -((y[0] * eye[0] + y[1] * eye[1]) + y[2] * eye[2])
    m[2, 3] = -((z[0] * eye[0] + z[1] * eye[1]) + z[2] * eye[2])
    return m


TAN_35DEG = 0.7002075382097097  # tan(70deg / 2), literal so no libm dependence


def perspective(tan_half_fov: float, aspect: float, znear: float, zfar: float) -> np.ndarray:
    """init_perspective (our_gl.cpp:44-56) with tan(fov/2) passed in."""
    m = np.eye(4)
    m[0, 0] = 1.0 / (aspect * tan_half_fov)
    m[1, 1] = 1.0 / tan_half_fov
    m[2, 2] = (zfar + znear) / (znear - zfar)
    m[2, 3] = (2.0 * zfar * znear) / (znear - zfar)
    m[3, 2] = -1.0
    m[3, 3] = 0.0
    return m


def head_standin(level: int, width: int, height: int, seed: int = SEED_DEFAULT, distance: float = 2.6):
    """Displaced icosphere through the vertex stage.  Returns dict with clip, varyings (K=24),
    model_view, light dirs (eye space, main.cpp:55-69 with the lights of main.cpp:615-617)."""
    tris = icosphere(level)                      # [F,3,3]
    n_unit = tris
    # smooth radial displacement from a few low-frequency lobes (polynomial, no trig)
    rng = SplitMix64(seed ^ 0xA5A5)
    lobes = _normalize_rows(rng.uniform(8 * 3, -1.0, 1.0).reshape(8, 3))
    amp = rng.uniform(8, 0.02, 0.08)
    disp = np.ones(tris.shape[:2])
    for l, a in zip(lobes, amp):
        d = (n_unit[..., 0] * l[0] + n_unit[..., 1] * l[1]) + n_unit[..., 2] * l[2]
        disp = disp + a * d * d * d
    pos = n_unit * disp[..., None]
    nrm = n_unit                                  # analytic-ish normal: undisplaced direction
    uv = _oct_uv(n_unit)

    mv = lookat([distance * 0.6, distance * 0.35, distance * 0.75], [0, 0, 0], [0, 1, 0])
    proj = perspective(TAN_35DEG, width / height, 0.05, 500.0)   # main.cpp:592-594
    ex, ey, ez, ew = _matvec(mv, pos[..., 0], pos[..., 1], pos[..., 2], 1.0)
    nx, ny, nz, _ = _matvec(mv, nrm[..., 0], nrm[..., 1], nrm[..., 2], 0.0)
    cx, cy, cz, cw = _matvec(proj, ex, ey, ez, ew)
    F = tris.shape[0]
    clip = np.stack([cx, cy, cz, cw], -1).reshape(F, 12)
    varyings = np.concatenate([uv.reshape(F, 6), np.stack([ex, ey, ez], -1).reshape(F, 9),
                               np.stack([nx, ny, nz], -1).reshape(F, 9)], 1)

    def light(d):
        d = np.asarray(d, np.float64)
        d = _normalize_rows(d)                                     # main.cpp:615-617
        e = np.array([((0.0 + mv[r, 0] * d[0]) + mv[r, 1] * d[1]) + mv[r, 2] * d[2] for r in range(3)])
        return _normalize_rows(e)                                  # main.cpp:59-68
    world = {k: _normalize_rows(np.asarray(v, np.float64)) for k, v in
             dict(key=[1.0, 1.4, 1.0], fill=[-0.3, 0.5, 0.2], rim=[-1.0, 0.8, -1.5]).items()}
    return dict(clip=np.ascontiguousarray(clip), varyings=np.ascontiguousarray(varyings), model_view=mv, projection=proj,
                positions=np.ascontiguousarray(pos), normals=np.ascontiguousarray(nrm), uvs=np.ascontiguousarray(uv),
                world_lights=world,
                key=light([1.0, 1.4, 1.0]), fill=light([-0.3, 0.5, 0.2]), rim=light([-1.0, 0.8, -1.5]))


# --------------------------------------------------------------------------------------------
# Procedural textures (integer lattice noise, bilinear in integer arithmetic): B,G,R[,A] bytes in
# TGAImage::buffer() layout.
# --------------------------------------------------------------------------------------------
def _lattice_noise(size: int, cell: int, seed: int) -> np.ndarray:
    g = size // cell + 2
    lat = (SplitMix64(seed).u64(g * g) >> np.uint64(56)).astype(np.int64).reshape(g, g)  # 0..255
    y, x = np.mgrid[0:size, 0:size]
    gx, gy, fx, fy = x // cell, y // cell, x % cell, y % cell
    v00, v10, v01, v11 = lat[gy, gx], lat[gy, gx + 1], lat[gy + 1, gx], lat[gy + 1, gx + 1]
    top = v00 * (cell - fx) + v10 * fx
    bot = v01 * (cell - fx) + v11 * fx
    return ((top * (cell - fy) + bot * fy) // (cell * cell)).astype(np.uint8)


def procedural_textures(size: int = 1024, seed: int = SEED_DEFAULT):
    """diffuse (BGR), normal (BGR, object-space style), specular (1 channel) maps."""
    n1 = _lattice_noise(size, 64, seed + 1).astype(np.int64)
    n2 = _lattice_noise(size, 16, seed + 2).astype(np.int64)
    n3 = _lattice_noise(size, 32, seed + 3).astype(np.int64)
    y, x = np.mgrid[0:size, 0:size]
    checker = (((x // 64) + (y // 64)) & 1) * 40
    diffuse = np.stack([np.clip(60 + n1 // 2 + checker, 0, 255), np.clip(80 + n2 // 2 + checker, 0, 255),
                        np.clip(120 + n3 // 2 + checker, 0, 255)], -1).astype(np.uint8)
    # a few near-white patches so PhongShader's is_eye_pixel branch (main.cpp:110-112) is exercised
    white = (n1 > 200)
    diffuse[white] = (250, 248, 245)
    normal = np.stack([np.clip(200 + n2 // 5, 0, 255), np.clip(96 + n3 // 4, 0, 255),
                       np.clip(96 + n1 // 4, 0, 255)], -1).astype(np.uint8)   # B=z, G=y, R=x
    specular = n2.astype(np.uint8)[..., None]
    return np.ascontiguousarray(diffuse), np.ascontiguousarray(normal), np.ascontiguousarray(specular)


# --------------------------------------------------------------------------------------------
# Edge cases the reference's setup code branches on (our_gl.cpp:94-135), used by parity tests.
# --------------------------------------------------------------------------------------------
def edge_case_triangles(width: int, height: int, seed: int = 7):
    """A few hundred triangles hitting every reject branch plus ordinary ones in between."""
    base, _ = random_triangles(256, width, height, seed=seed, rmin=4.0, rmax=64.0)
    c = base.copy()
    inf, nan = np.inf, np.nan
    c[3, 3] = 0.0                      # w == 0            -> :94
    c[5, 7] = -1.0                     # negative w        -> :94
    c[7, 11] = 1e-13                   # w <= 1e-12        -> :94
    c[9, [2, 6, 10]] = [1.5, -1.2, 2.0]    # every vertex outside z range -> :103-106
    c[11, [2, 6, 10]] = [1.5, 0.0, -2.0]   # partially outside: still drawn
    c[13, 0] = nan                     # NaN x             -> :109-114
    c[15, 5] = inf                     # inf y             -> :109-114
    c[17] = c[17][[0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7]]   # swapped winding -> back face :127
    c[19, 4:8] = c[19, 0:4]            # degenerate (two equal vertices) -> cross == 0
    c[21, [0, 4, 8]] += 5.0            # entirely right of the screen -> empty bbox :135
    c[23, [1, 5, 9]] -= 5.0            # entirely below
    c[25, 3] = 1e-11; c[25, 0] = 1.0   # tiny positive w -> |ndc| ~ 1e11 -> (int) cast out of range :131
    c[27, [0, 1]] = [-30.0, -30.0]     # huge on-screen triangle, vertex far outside
    c[27, [4, 5]] = [30.0, -30.0]
    c[27, [8, 9]] = [0.0, 30.0]
    c[27, [2, 6, 10]] = [0.9, 0.9, 0.9]
    c[29] = c[28]                      # exact duplicate: z tie, earlier triangle must win :165
    c[31, [2, 6, 10]] = [-1.0, 1.0, 0.0]   # z exactly on the range ends
    c[33, 3] = 2.0; c[33, 7] = 0.5; c[33, 11] = 1.25   # mixed w on an otherwise ordinary triangle
    colors = pack_bgra((np.arange(256) * 7) & 255, (np.arange(256) * 13) & 255, (np.arange(256) * 29) & 255)
    return c, colors


def shared_edge_grid(nx: int, ny: int, width: int, height: int, z_slope: float = 0.25):
    """A regular grid of quads split into triangles: every interior edge is shared, so edge pixels
    are covered by both neighbours (`>= 0`, our_gl.cpp:152) and the earlier triangle must win the
    z tie (strict `<`, our_gl.cpp:165).  Vertices sit on exact pixel centres on purpose."""
    xs = (np.arange(nx + 1) * (width // nx) + 0.5) * (2.0 / width) - 1.0
    ys = (np.arange(ny + 1) * (height // ny) + 0.5) * (2.0 / height) - 1.0
    tris = []
    for j in range(ny):
        for i in range(nx):
            x0, x1, y0, y1 = xs[i], xs[i + 1], ys[j], ys[j + 1]
            z = lambda x, y: z_slope * x - z_slope * y
            tris.append([x0, y0, z(x0, y0), 1, x1, y0, z(x1, y0), 1, x1, y1, z(x1, y1), 1])
            tris.append([x0, y0, z(x0, y0), 1, x1, y1, z(x1, y1), 1, x0, y1, z(x0, y1), 1])
    clip = np.array(tris, dtype=np.float64)
    n = clip.shape[0]
    colors = pack_bgra((np.arange(n) * 37) & 255, (np.arange(n) * 91) & 255, (np.arange(n) * 53) & 255)
    return clip, colors
