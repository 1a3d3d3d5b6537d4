"""Python host mirror of the C ABI in include/trgl.h (ctypes over tinyrenderder_amd/libtrgl.so).

This is the product path: it never falls back to a CPU implementation.  If the HIP library is
missing or a call fails, it raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libtrgl.so")

FLAT, GOURAUD, PHONG, EYE, CHECKER = 0, 1, 2, 3, 4
VARY = {FLAT: 0, GOURAUD: 3, PHONG: 24, EYE: 24, CHECKER: 0}
MEM_HOST, MEM_DEVICE = 0, 1
PHASE_SETUP, PHASE_BIN, PHASE_RASTER, PHASE_TOTAL, PHASE_RASTER_KERNEL = 0, 1, 2, 3, 4
NUM_PHASES = 5        # TRGL_NUM_PHASES
MAX_TEXTURES = 16

# every symbol include/trgl.h declares (tests check the library exports all of them)
SYMBOLS = [
    "trgl_create", "trgl_destroy", "trgl_last_error", "trgl_set_viewport", "trgl_init_viewport", "trgl_clear",
    "trgl_upload_texture", "trgl_set_strip", "trgl_set_interleave", "trgl_draw", "trgl_flush", "trgl_flush_begin", "trgl_flush_end", "trgl_sync", "trgl_read_framebuffer",
    "trgl_write_framebuffer", "trgl_read_zbuffer", "trgl_write_zbuffer", "trgl_get_stats", "trgl_reset_stats",
    "trgl_format_stats", "trgl_framebuffer_device_ptr", "trgl_zbuffer_device_ptr", "trgl_stream", "trgl_set_stream",
    "trgl_set_profiling", "trgl_get_phase_ms", "trgl_reset_phase_ms", "trgl_get_last_flush_info",
    "trgl_selftest_division", "trgl_selftest_sampler", "trgl_tga_max_size", "trgl_tga_encode", "trgl_tga_info", "trgl_tga_decode", "trgl_draw_indexed", "trgl_ssao_defaults",
    "trgl_postprocess", "trgl_obj_load", "trgl_obj_free",
    "trgl_gather", "trgl_rccl_unique_id", "trgl_rccl_comm_create", "trgl_rccl_comm_destroy",
]


class Uniforms(C.Structure):
    """trgl_uniforms"""
    _fields_ = [("model_view", C.c_double * 16), ("key_light_dir_eye", C.c_double * 3),
                ("fill_light_dir_eye", C.c_double * 3), ("rim_light_dir_eye", C.c_double * 3),
                ("normal_map_strength", C.c_double), ("tex_diffuse", C.c_int32), ("tex_normal", C.c_int32),
                ("tex_specular", C.c_int32), ("reserved", C.c_int32)]


class Stats(C.Structure):
    """trgl_stats"""
    _fields_ = [("triangles_rasterized", C.c_uint64), ("fragments_drawn", C.c_uint64),
                ("min_x", C.c_int32), ("min_y", C.c_int32), ("max_x", C.c_int32), ("max_y", C.c_int32),
                ("min_z", C.c_double), ("max_z", C.c_double)]

    def astuple(self):
        # z range as (value, sign bit) so that -0.0 and +0.0 compare unequal, as their printed forms do
        import math
        return (self.triangles_rasterized, self.fragments_drawn, self.min_x, self.min_y, self.max_x, self.max_y,
                self.min_z, self.max_z, math.copysign(1.0, self.min_z), math.copysign(1.0, self.max_z))


class SsaoParams(C.Structure):
    """trgl_ssao_params"""
    _fields_ = [("num_directions", C.c_int32), ("steps_per_direction", C.c_int32), ("sample_radius", C.c_double),
                ("occlusion_threshold", C.c_double), ("intensity", C.c_double)]


def make_uniforms(model_view=None, key=(0, 0, 1), fill=(0, 0, 1), rim=(0, 0, 1), normal_map_strength=1.0,
                  tex_diffuse=-1, tex_normal=-1, tex_specular=-1, cells=0) -> Uniforms:
    u = Uniforms()
    mv = np.eye(4) if model_view is None else np.asarray(model_view, np.float64)
    u.model_view[:] = mv.reshape(16).tolist()
    u.key_light_dir_eye[:] = list(map(float, key))
    u.fill_light_dir_eye[:] = list(map(float, fill))
    u.rim_light_dir_eye[:] = list(map(float, rim))
    u.normal_map_strength = float(normal_map_strength)
    u.tex_diffuse, u.tex_normal, u.tex_specular, u.reserved = tex_diffuse, tex_normal, tex_specular, int(cells)      # cells: CHECKER only
    return u


class TrglError(RuntimeError):
    pass


_lib = None


def load_library(path: str = None):
    """Load libtrgl.so and declare every prototype.  Raises if the library is absent.
    TRGL_LIB in the environment names another build of the same library (profiles/: diagnostic and A/B builds)."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("TRGL_LIB") or LIB_PATH
    # One HIP runtime per process: torch wheels bundle their own libamdhip64.  If libtrgl.so pulled in
    # /opt/rocm's copy first, a later `import torch` in the same process finds no GPUs; loaded after torch,
    # libtrgl.so binds to the runtime that is already there.  (A C/C++ host without torch is unaffected.)
    if "torch" not in sys.modules and not os.environ.get("TRGL_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(path):
        raise TrglError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(path)
    vp, u64p, dp = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)
    L.trgl_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.trgl_destroy.argtypes = [vp]
    L.trgl_last_error.argtypes = [vp]; L.trgl_last_error.restype = C.c_char_p
    L.trgl_set_viewport.argtypes = [vp, dp]
    L.trgl_init_viewport.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.trgl_clear.argtypes = [vp, C.c_void_p, C.c_double]
    L.trgl_upload_texture.argtypes = [vp, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.trgl_set_strip.argtypes = [vp, C.c_int, C.c_int]
    L.trgl_set_interleave.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.trgl_draw.argtypes = [vp, C.c_int, C.POINTER(Uniforms), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
    L.trgl_flush.argtypes = [vp]
    L.trgl_flush_begin.argtypes = [vp]
    L.trgl_flush_end.argtypes = [vp]
    L.trgl_sync.argtypes = [vp]
    L.trgl_read_framebuffer.argtypes = [vp, C.c_void_p]
    L.trgl_write_framebuffer.argtypes = [vp, C.c_void_p]
    L.trgl_read_zbuffer.argtypes = [vp, C.c_void_p]
    L.trgl_write_zbuffer.argtypes = [vp, C.c_void_p]
    L.trgl_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.trgl_reset_stats.argtypes = [vp]
    L.trgl_format_stats.argtypes = [C.POINTER(Stats), C.c_char_p, C.c_size_t]
    for name in ("trgl_framebuffer_device_ptr", "trgl_zbuffer_device_ptr", "trgl_stream"):
        getattr(L, name).argtypes = [vp]; getattr(L, name).restype = C.c_void_p
    L.trgl_set_stream.argtypes = [vp, C.c_void_p, C.c_int]
    L.trgl_set_profiling.argtypes = [vp, C.c_int]
    L.trgl_get_phase_ms.argtypes = [vp, dp, u64p]
    L.trgl_reset_phase_ms.argtypes = [vp]
    L.trgl_get_last_flush_info.argtypes = [vp, u64p, u64p, u64p]
    L.trgl_selftest_division.argtypes = [vp, C.c_uint64, C.c_uint64, u64p]
    L.trgl_selftest_sampler.argtypes = [vp, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]
    L.trgl_draw_indexed.argtypes = [vp, C.c_int, C.POINTER(Uniforms), dp, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int]
    L.trgl_ssao_defaults.argtypes = [C.POINTER(SsaoParams)]
    L.trgl_ssao_defaults.restype = None
    L.trgl_postprocess.argtypes = [vp, C.POINTER(SsaoParams), C.c_void_p, C.c_void_p, C.c_void_p]
    L.trgl_obj_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_double)), u64p, C.POINTER(C.POINTER(C.c_uint32)), u64p]
    L.trgl_obj_free.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    L.trgl_obj_free.restype = None
    L.trgl_tga_max_size.argtypes = [C.c_int, C.c_int, C.c_int]
    L.trgl_tga_max_size.restype = C.c_size_t
    L.trgl_tga_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_size_t)]
    L.trgl_tga_info.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.trgl_tga_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.trgl_gather.argtypes = [vp, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.trgl_rccl_unique_id.argtypes = [C.c_void_p]
    L.trgl_rccl_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.trgl_rccl_comm_destroy.argtypes = [C.c_void_p]
    for name in SYMBOLS:
        f = getattr(L, name)
        if f.restype is C.c_int and name not in ("trgl_last_error",):
            f.restype = C.c_int
    _lib = L
    return L


def rccl_unique_id() -> bytes:
    """ncclGetUniqueId through the C ABI: 128 bytes that rank 0 hands to the other ranks."""
    L = load_library()
    buf = (C.c_uint8 * 128)()
    rc = L.trgl_rccl_unique_id(buf)
    if rc != 0:
        raise TrglError(f"trgl_rccl_unique_id failed ({rc}): {L.trgl_last_error(None).decode()}")
    return bytes(buf)


def rccl_comm_create(unique_id: bytes, rank: int, world: int, device: int = 0):
    """ncclCommInitRank through the C ABI; returns the communicator handle for Context.gather()."""
    L = load_library()
    comm = C.c_void_p()
    buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
    rc = L.trgl_rccl_comm_create(buf, rank, world, device, C.byref(comm))
    if rc != 0:
        raise TrglError(f"trgl_rccl_comm_create failed ({rc}): {L.trgl_last_error(None).decode()}")
    return comm


def rccl_comm_destroy(comm):
    load_library().trgl_rccl_comm_destroy(comm)


def _ptr(a):
    """host numpy array, or an object with a device pointer (torch tensor / int)."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return a.ctypes.data


def tga_encode(img, vflip: bool = True, rle: bool = True) -> bytes:
    """The bytes the reference's TGAImage::write_tga_file would write for an [h,w,bpp] uint8 image (host only)."""
    L = load_library()
    img = np.ascontiguousarray(img, np.uint8)
    if img.ndim == 2:
        img = img[..., None]
    h, w, bpp = img.shape
    out = np.empty(L.trgl_tga_max_size(w, h, bpp), np.uint8)
    n = C.c_size_t()
    rc = L.trgl_tga_encode(img.ctypes.data, w, h, bpp, int(vflip), int(rle), out.ctypes.data, C.byref(n))
    if rc != 0:
        raise TrglError(f"trgl_tga_encode failed ({rc})")
    return out[:n.value].tobytes()


def tga_decode(file_bytes: bytes):
    """TGAImage::read_tga_file on a .tga file image: [h,w,bpp] uint8 in TGAImage::buffer() order (host only).
    Raises TrglError where the reference returns false."""
    L = load_library()
    buf = np.frombuffer(file_bytes, np.uint8)
    w, h, bpp = C.c_int(), C.c_int(), C.c_int()
    rc = L.trgl_tga_info(buf.ctypes.data if len(buf) else None, len(buf), C.byref(w), C.byref(h), C.byref(bpp))
    if rc != 0:
        raise TrglError(f"trgl_tga_info: not a TGA file the reference reads ({rc})")
    out = np.empty((h.value, w.value, bpp.value), np.uint8)
    rc = L.trgl_tga_decode(buf.ctypes.data, len(buf), out.ctypes.data)
    if rc != 0:
        raise TrglError(f"trgl_tga_decode failed ({rc})")
    return out


def load_obj(path: str):
    """Wavefront OBJ -> (vertices [nv,14] f64 in the reference's Vertex layout, indices [nf,3] u32).  Host only."""
    L = load_library()
    v, i = C.POINTER(C.c_double)(), C.POINTER(C.c_uint32)()
    nv, nf = C.c_uint64(), C.c_uint64()
    rc = L.trgl_obj_load(path.encode(), C.byref(v), C.byref(nv), C.byref(i), C.byref(nf))
    if rc != 0:
        raise TrglError(f"trgl_obj_load failed ({rc}): {L.trgl_last_error(None).decode()}")
    try:
        verts = np.ctypeslib.as_array(v, shape=(nv.value, 14)).copy() if nv.value else np.zeros((0, 14))
        idx = np.ctypeslib.as_array(i, shape=(nf.value, 3)).copy() if nf.value else np.zeros((0, 3), np.uint32)
    finally:
        L.trgl_obj_free(v, i)
    return verts, idx


class Context:
    """One rasterizer context on one GPU — the reference's globals (Viewport, zbuffer, counters,
    our_gl.cpp:12-22) plus the framebuffer TGAImage, as an object."""

    def __init__(self, width: int, height: int, bpp: int = 3, device: int = 0):
        self.L = load_library()
        self.h = C.c_void_p()
        rc = self.L.trgl_create(device, width, height, bpp, C.byref(self.h))
        if rc != 0:
            raise TrglError(f"trgl_create failed ({rc}): {self.L.trgl_last_error(None).decode()}")
        self.width, self.height, self.bpp, self.device = width, height, bpp, device
        self._keep = []

    def _chk(self, rc):
        if rc != 0:
            raise TrglError(f"trgl call failed ({rc}): {self.L.trgl_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.L.trgl_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- state ----
    def set_viewport(self, m):
        m = np.ascontiguousarray(m, np.float64).reshape(16)
        self._chk(self.L.trgl_set_viewport(self.h, m.ctypes.data_as(C.POINTER(C.c_double))))

    def init_viewport(self, x, y, w, h):
        self._chk(self.L.trgl_init_viewport(self.h, x, y, w, h))

    def clear(self, bgra=None, z=np.inf):
        b = None if bgra is None else np.asarray(bgra, np.uint8)
        self._chk(self.L.trgl_clear(self.h, None if b is None else b.ctypes.data, float(z)))

    def upload_texture(self, slot, texels):
        t = np.ascontiguousarray(texels, np.uint8)
        if t.ndim == 2:
            t = t[..., None]
        self._chk(self.L.trgl_upload_texture(self.h, slot, t.ctypes.data, t.shape[1], t.shape[0], t.shape[2]))

    def set_strip(self, y0, y1):
        self._chk(self.L.trgl_set_strip(self.h, y0, y1))

    def set_interleave(self, band_rows, rank, world):
        """Own the bands of `band_rows` rows whose number is `rank` modulo `world` (instead of one strip)."""
        self._chk(self.L.trgl_set_interleave(self.h, band_rows, rank, world))

    def gather(self, comm, rank, world, with_z=False):
        """trgl_gather: join the rows of the `world` contexts of a render into this context's framebuffer (and z-buffer) with
        in-place RCCL all-gathers queued on the context's stream; `comm` is an ncclComm_t (rccl_comm_create)."""
        self._chk(self.L.trgl_gather(self.h, comm, rank, world, int(bool(with_z))))

    # ---- submission ----
    def draw(self, kind, clip, varyings=None, colors=None, uniforms=None, n=None, device=False):
        """Host arrays (numpy) are copied before return; with device=True pass torch CUDA tensors (or raw
        pointers with n) that stay alive until the flush has completed."""
        K = VARY[kind]
        if not device:
            clip = np.ascontiguousarray(clip, np.float64)
            n = clip.shape[0] if n is None else n
            if K:
                varyings = np.ascontiguousarray(varyings, np.float64)
                assert varyings.shape == (n, K), varyings.shape
            if colors is not None:
                colors = np.ascontiguousarray(colors, np.uint32)
                assert colors.shape == (n,)
        else:
            assert n is not None or hasattr(clip, "shape")
            n = clip.shape[0] if n is None else n
            self._keep.append((clip, varyings, colors))
        self._chk(self.L.trgl_draw(self.h, kind, None if uniforms is None else C.byref(uniforms), _ptr(clip),
                                   _ptr(varyings) if K else None, _ptr(colors), int(n), MEM_DEVICE if device else MEM_HOST))

    def draw_indexed(self, kind, uniforms, projection, vertices, indices, device=False):
        """Vertex stage on the device (main.cpp:71-90) + draw.  vertices [nv, stride>=8] f64, indices [nf,3] u32."""
        pj = np.ascontiguousarray(projection, np.float64).reshape(16)
        if not device:
            vertices = np.ascontiguousarray(vertices, np.float64)
            indices = np.ascontiguousarray(indices, np.uint32).reshape(-1, 3)
        else:
            self._keep.append((vertices, indices))
        nv, stride = vertices.shape
        nf = indices.shape[0]
        self._chk(self.L.trgl_draw_indexed(self.h, kind, C.byref(uniforms), pj.ctypes.data_as(C.POINTER(C.c_double)), _ptr(vertices),
                                           stride, nv, _ptr(indices), nf, MEM_DEVICE if device else MEM_HOST))

    def postprocess(self, zbuffer_image=True, ao=True, final=True, params=None):
        """main.cpp:269-311,317-362,757-783 on the device; returns dict of [h,w,3] uint8 images."""
        out = {}
        shape = (self.height, self.width, 3)
        zi = np.empty(shape, np.uint8) if zbuffer_image else None
        a = np.empty(shape, np.uint8) if ao else None
        f = np.empty(shape, np.uint8) if final else None
        self._chk(self.L.trgl_postprocess(self.h, None if params is None else C.byref(params),
                                          None if zi is None else zi.ctypes.data, None if a is None else a.ctypes.data,
                                          None if f is None else f.ctypes.data))
        return dict(zbuffer_image=zi, ao=a, final=f)

    def flush(self):
        self._chk(self.L.trgl_flush(self.h))

    def flush_begin(self):
        """Setup + binning of what was submitted (does not touch the framebuffer / z-buffer); flush_end() runs the raster."""
        self._chk(self.L.trgl_flush_begin(self.h))

    def flush_end(self):
        self._chk(self.L.trgl_flush_end(self.h))

    def sync(self):
        self._chk(self.L.trgl_sync(self.h))
        self._keep.clear()

    # ---- results ----
    def read_framebuffer(self) -> np.ndarray:
        out = np.empty((self.height, self.width, self.bpp), np.uint8)
        self._chk(self.L.trgl_read_framebuffer(self.h, out.ctypes.data))
        self._keep.clear()
        return out

    def read_zbuffer(self) -> np.ndarray:
        out = np.empty((self.height, self.width), np.float64)
        self._chk(self.L.trgl_read_zbuffer(self.h, out.ctypes.data))
        self._keep.clear()
        return out

    def write_framebuffer(self, fb):
        fb = np.ascontiguousarray(fb, np.uint8)
        assert fb.size == self.width * self.height * self.bpp
        self._chk(self.L.trgl_write_framebuffer(self.h, fb.ctypes.data))

    def write_zbuffer(self, z):
        z = np.ascontiguousarray(z, np.float64)
        assert z.size == self.width * self.height
        self._chk(self.L.trgl_write_zbuffer(self.h, z.ctypes.data))

    def stats(self):
        s = Stats()
        self._chk(self.L.trgl_get_stats(self.h, C.byref(s)))
        return s.astuple()

    def stats_line(self) -> str:
        s = Stats()
        self._chk(self.L.trgl_get_stats(self.h, C.byref(s)))
        buf = C.create_string_buffer(1024)
        self._chk(self.L.trgl_format_stats(C.byref(s), buf, 1024))
        return buf.value.decode().strip()

    def reset_stats(self):
        self._chk(self.L.trgl_reset_stats(self.h))

    @property
    def framebuffer_ptr(self) -> int:
        return self.L.trgl_framebuffer_device_ptr(self.h)

    @property
    def zbuffer_ptr(self) -> int:
        return self.L.trgl_zbuffer_device_ptr(self.h)

    @property
    def stream(self) -> int:
        return self.L.trgl_stream(self.h)

    def set_stream(self, hip_stream, use_own: bool = False):
        """Enqueue on the caller's hipStream_t (int handle, e.g. torch.cuda.current_stream().cuda_stream; 0 is the
        legacy default stream).  use_own=True returns to the context's own stream."""
        self._chk(self.L.trgl_set_stream(self.h, hip_stream or None, 1 if use_own else 0))

    # ---- measurement ----
    def set_profiling(self, on: bool):
        self._chk(self.L.trgl_set_profiling(self.h, 1 if on else 0))

    def phase_ms(self):
        ms = (C.c_double * NUM_PHASES)()
        n = C.c_uint64()
        self._chk(self.L.trgl_get_phase_ms(self.h, ms, C.byref(n)))
        return list(ms), n.value

    def reset_phase_ms(self):
        self._chk(self.L.trgl_reset_phase_ms(self.h))

    def selftest_division(self, samples: int, seed: int = 1) -> int:
        bad = C.c_uint64()
        self._chk(self.L.trgl_selftest_division(self.h, samples, seed, C.byref(bad)))
        return bad.value

    def selftest_sampler(self, slot: int, uv) -> np.ndarray:
        """The device samplers' texel fetch of texture `slot` at uv [n,2]: [n,5] uint8 = bgra[4], bytespp."""
        uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
        out = np.zeros((uv.shape[0], 5), np.uint8)
        self._chk(self.L.trgl_selftest_sampler(self.h, slot, uv.ctypes.data, uv.shape[0], out.ctypes.data))
        return out

    def last_flush_info(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(self.L.trgl_get_last_flush_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(triangles=a.value, pairs=b.value, tiles=c.value)
