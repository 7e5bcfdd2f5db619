"""CPU oracle for the path tracer's hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package; the product (``raytracer_amd``) never
does.  ``oracle_rt.c`` is the restatement of the reference algorithm,
``ref_driver.c`` wraps the reference's own kd-tree builder where the reference
checkout is available (output in ``oracle/_ref/``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

HIT_DT = np.dtype([("tri", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])


class KdStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("leaves_visited", C.c_uint64),
                ("tri_refs", C.c_uint64), ("tri_tests", C.c_uint64), ("pushes", C.c_uint64),
                ("rays_over_16", C.c_uint64), ("max_stack", C.c_uint32), ("pad", C.c_uint32),
                ("empty_leaves", C.c_uint64)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "pad"}


def build(force: bool = False) -> None:
    """Compile liboracle_rt.so (and _ref/libref_kdtree.so when the reference
    checkout is mounted)."""
    if force or not os.path.exists(os.path.join(_HERE, "liboracle_rt.so")) or \
            os.path.getmtime(os.path.join(_HERE, "liboracle_rt.so")) < os.path.getmtime(os.path.join(_HERE, "oracle_rt.c")):
        subprocess.check_call(["make", "-C", _HERE, "liboracle_rt.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src") and (force or not os.path.exists(os.path.join(_HERE, "_ref", "libref_kdtree.so"))):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle_rt.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_kd_build.restype = C.c_void_p
        _LIB.orc_kd_blob_size.restype = C.c_uint32
        _LIB.orc_kd_serialize.restype = C.c_uint32
        _LIB.orc_max_threads.restype = C.c_int
    return _LIB


def ref_lib():
    """The reference's own builder, or None where it has not been built."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libref_kdtree.so")
        if not os.path.exists(path):
            return None
        _REF = C.CDLL(path)
        _REF.ref_kd_build.restype = C.c_uint
    return _REF


_REF_OBJ = None


class _RefObj(C.Structure):
    _fields_ = [("vertices", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)),
                ("texcoords", C.POINTER(C.c_float)),
                ("num_vertices", C.c_uint), ("num_normals", C.c_uint), ("num_texcoords", C.c_uint),
                ("faces", C.POINTER(C.c_int)), ("num_faces", C.c_uint),
                ("face_num_verts", C.POINTER(C.c_int)), ("num_face_num_verts", C.c_uint),
                ("shape_face_offset", C.POINTER(C.c_uint)), ("shape_length", C.POINTER(C.c_uint)),
                ("num_shapes", C.c_uint)]


def ref_obj_parse(data: bytes):
    """One OBJ file through the parser the reference vendors (oracle/_ref/libref_tinyobj.so), or None
    where it has not been built.  `data` is the file's bytes; the terminating NUL the reference's
    load_file appends is added here."""
    global _REF_OBJ
    if _REF_OBJ is None:
        path = os.path.join(_HERE, "_ref", "libref_tinyobj.so")
        if not os.path.exists(path):
            return None
        _REF_OBJ = C.CDLL(path)
        _REF_OBJ.ref_obj_parse.argtypes = [C.POINTER(_RefObj), C.c_char_p, C.c_size_t]
    buf = data + b"\0"
    o = _RefObj()
    rc = _REF_OBJ.ref_obj_parse(C.byref(o), buf, len(buf))
    if rc != 0:
        raise RuntimeError(f"reference OBJ parser returned {rc}")

    def arr(ptr, n, dt):
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)
    return {
        "vertices": arr(o.vertices, 3 * o.num_vertices, np.float32).reshape(-1, 3),
        "normals": arr(o.normals, 3 * o.num_normals, np.float32).reshape(-1, 3),
        "texcoords": arr(o.texcoords, 2 * o.num_texcoords, np.float32).reshape(-1, 2),
        "faces": arr(o.faces, 3 * o.num_faces, np.int32).reshape(-1, 3),          # {v, vt, vn} per corner
        "face_num_verts": arr(o.face_num_verts, o.num_face_num_verts, np.int32),
        "shape_face_offset": arr(o.shape_face_offset, o.num_shapes, np.uint32),
        "shape_length": arr(o.shape_length, o.num_shapes, np.uint32),
    }


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def set_builtins(mode: str) -> None:
    """'rocm' (default): dot/cross/min/max as ROCm's OpenCL library defines them on gfx950 -- what the reference's
    kernels compute when built for this GPU, and what the library is built to; 'plain': sums left to right, textbook
    cross, select-style min/max."""
    lib().orc_set_builtins(C.c_int({"plain": 0, "rocm": 1}[mode]))


def max_threads() -> int:
    return int(lib().orc_max_threads())


class KdTree:
    """Oracle kd-tree: pointer tree + the reference's serialised blob."""

    def __init__(self, verts4: np.ndarray, idx4: np.ndarray):
        self.verts4 = np.ascontiguousarray(verts4, np.float32)
        self.idx4 = np.ascontiguousarray(idx4, np.int32)
        L = lib()
        self._h = C.c_void_p(L.orc_kd_build(_p(self.verts4), C.c_uint32(self.verts4.shape[0]),
                                            _p(self.idx4), C.c_uint32(self.idx4.shape[0])))
        c = np.zeros(7, np.uint32)
        L.orc_kd_counts(self._h, _p(c))
        self.counts = dict(zip(["nodes_total", "traversal", "leaves", "indices_total", "tris_padded",
                                "max_depth", "count_mismatch"], (int(x) for x in c)))
        self.bounds_min = np.zeros(3, np.float32)
        self.bounds_max = np.zeros(3, np.float32)
        L.orc_kd_bounds(self._h, _p(self.bounds_min), _p(self.bounds_max))
        size = int(L.orc_kd_blob_size(self._h))
        self.blob = np.zeros(max(size, 8), np.uint8)
        self.blob_used = int(L.orc_kd_serialize(self._h, _p(self.blob)))
        self.blob_size = size
        self.scene_min = np.zeros(3, np.float32)
        self.scene_max = np.zeros(3, np.float32)
        L.orc_scene_bounds_rounded(_p(self.bounds_min), _p(self.bounds_max), _p(self.scene_min), _p(self.scene_max))

    @classmethod
    def from_blob(cls, blob: np.ndarray, verts4: np.ndarray, idx4: np.ndarray, bounds_min, bounds_max):
        """Oracle traversal over an already serialised tree (the reference's
        wire format), e.g. one a caller built elsewhere; no oracle build."""
        self = cls.__new__(cls)
        self.verts4 = np.ascontiguousarray(verts4, np.float32)
        self.idx4 = np.ascontiguousarray(idx4, np.int32)
        self._h = None
        self.blob = np.ascontiguousarray(blob, np.uint8)
        self.blob_size = self.blob_used = int(self.blob.shape[0])
        self.bounds_min = np.asarray(bounds_min, np.float32).copy()
        self.bounds_max = np.asarray(bounds_max, np.float32).copy()
        self.scene_min = np.zeros(3, np.float32)
        self.scene_max = np.zeros(3, np.float32)
        lib().orc_scene_bounds_rounded(_p(self.bounds_min), _p(self.bounds_max), _p(self.scene_min), _p(self.scene_max))
        self.counts = {}
        return self

    def __del__(self):
        try:
            if self._h is not None:
                lib().orc_kd_free(self._h)
        except Exception:
            pass

    def intersect(self, rays: np.ndarray, stats: bool = False, nthreads: int = 0):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(rays.shape[0], HIT_DT)
        st = KdStats()
        lib().orc_kd_intersect(_p(self.blob), _p(self.idx4), _p(self.verts4), _p(rays), C.c_uint32(rays.shape[0]),
                               _p(self.scene_min), _p(self.scene_max), _p(out),
                               C.byref(st) if stats else None, C.c_int(nthreads or max_threads()))
        return (out, st.as_dict()) if stats else out

    def ray_stack(self, rays: np.ndarray, nthreads: int = 0) -> np.ndarray:
        """Deepest traversal stack each ray needs (the reference's own stack holds 16 entries)."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        depth = np.zeros(rays.shape[0], np.uint32)
        lib().orc_kd_ray_stack(_p(self.blob), _p(self.idx4), _p(self.verts4), _p(rays), C.c_uint32(rays.shape[0]),
                               _p(self.scene_min), _p(self.scene_max), _p(depth), C.c_int(nthreads or max_threads()))
        return depth

    def ray_steps(self, rays: np.ndarray, nthreads: int = 0) -> np.ndarray:
        """[n, 2] array: interior nodes visited and triangles tested per ray."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        steps = np.zeros((rays.shape[0], 2), np.uint32)
        lib().orc_kd_ray_steps(_p(self.blob), _p(self.idx4), _p(self.verts4), _p(rays), C.c_uint32(rays.shape[0]),
                               _p(self.scene_min), _p(self.scene_max), _p(steps), C.c_int(nthreads or max_threads()))
        return steps


def ref_kd_blob(verts4: np.ndarray, idx4: np.ndarray):
    """(blob bytes, counts5, bounds6) from the reference's own builder."""
    R = ref_lib()
    if R is None:
        raise RuntimeError("oracle/_ref/libref_kdtree.so is not built")
    verts4 = np.ascontiguousarray(verts4, np.float32)
    idx4 = np.ascontiguousarray(idx4, np.int32)
    blob = C.c_void_p()
    counts = np.zeros(5, np.uint32)
    bounds = np.zeros(6, np.float32)
    size = R.ref_kd_build(_p(verts4), C.c_uint(verts4.shape[0]), _p(idx4), C.c_uint(idx4.shape[0]),
                          C.byref(blob), _p(counts), _p(bounds))
    data = np.ctypeslib.as_array(C.cast(blob, C.POINTER(C.c_uint8)), shape=(int(size),)).copy()
    R.ref_kd_release(blob)
    return data, counts, bounds


def blob_significant_mask(blob: np.ndarray, used: int) -> np.ndarray:
    """Boolean mask of the bytes of a serialised tree that carry data (type,
    axis, split, child words, leaf counts, triangle offsets); the rest is
    struct padding the reference never initialises."""
    mask = np.zeros(blob.shape[0], bool)
    off = 0
    words = blob.view(np.uint8)
    while off < used:
        t = int(words[off])
        if t == 2:
            mask[off:off + 2] = True
            mask[off + 4:off + 24] = True
            off += 32
        elif t == 1:
            n = int(blob[off + 4:off + 8].view(np.uint32)[0])
            mask[off] = True
            mask[off + 4:off + 8] = True
            mask[off + 16:off + 16 + 4 * n] = True
            off += 16 + 4 * n + (4 if n % 2 else 0)
        else:
            raise ValueError(f"bad node type {t} at {off}")
    return mask


def generate_rays(w: int, h: int, y0: int = 0, full_h: int | None = None) -> np.ndarray:
    rays = np.zeros((w * h, 8), np.float32)
    lib().orc_generate_rays(_p(rays), C.c_uint32(w), C.c_uint32(h), C.c_uint32(y0), C.c_uint32(full_h or h))
    return rays


def brute_nearest(scene, rays: np.ndarray, nthreads: int = 0) -> np.ndarray:
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
    out = np.zeros(rays.shape[0], HIT_DT)
    lib().orc_brute_nearest(_p(scene.idx4), _p(scene.verts4), C.c_uint32(scene.idx4.shape[0]), _p(rays),
                            C.c_uint32(rays.shape[0]), _p(out), C.c_int(nthreads or max_threads()))
    return out


def sincos(theta: np.ndarray):
    theta = np.ascontiguousarray(theta, np.float32)
    s = np.zeros_like(theta)
    c = np.zeros_like(theta)
    lib().orc_sincos_array(_p(theta), _p(s), _p(c), C.c_uint32(theta.shape[0]))
    return s, c


def xorshift(data: np.ndarray) -> None:
    lib().orc_xorshift(_p(data), C.c_uint32(data.shape[0]))


class SpathState:
    """Host-side mirror of the split path tracer's per-pixel buffers
    (src/spath_raytracer.c:49-113) driven by the oracle kernels in the order of
    spath_raytracer_prepass / _render (src/spath_raytracer.c:355-436)."""

    def __init__(self, scene, tree: KdTree, w: int, h: int, y0: int = 0, full_h: int | None = None):
        self.scene, self.tree, self.w, self.h, self.y0 = scene, tree, w, h, y0
        self.full_h = full_h or h
        n = w * h
        self.n = n
        self.out_tex = np.zeros((n, 4), np.float32)
        self.rays = np.zeros((n, 8), np.float32)
        self.origin_rays = np.zeros((n, 8), np.float32)
        self.results = np.zeros(n, HIT_DT)
        self.source_results = np.zeros(n, HIT_DT)
        self.progress = np.zeros((n, 12), np.uint32)
        self.random = np.zeros(n, np.uint32)
        self.iteration = 0
        self.nthreads = max_threads()

    def prepass(self, random_buffer: np.ndarray, rv_init: int, zero_rays: bool = True):
        s = self.scene
        self.out_tex[:] = 0
        self.progress[:] = 0
        self.iteration = 0
        self.rays = generate_rays(self.w, self.h, self.y0, self.full_h)
        self.results = self.tree.intersect(self.rays, nthreads=self.nthreads)
        self.primary_rays = self.rays.copy()
        lib().orc_spath_init(_p(self.out_tex), _p(self.rays), _p(self.origin_rays), _p(self.results),
                             _p(self.source_results), _p(self.progress), _p(s.idx4), _p(s.normals4),
                             _p(s.meshes), _p(s.materials), C.c_uint32(self.w), C.c_uint32(self.h),
                             C.c_uint32(self.y0), C.c_uint32(rv_init))
        self.random = np.ascontiguousarray(random_buffer, np.uint32).copy()
        self.first_bounce_rays = self.rays.copy()
        if zero_rays:                      # src/spath_raytracer.c:432-433
            self.rays[:] = 0

    def iterate(self, rv: int):
        s = self.scene
        xorshift(self.random)
        self.iteration += 1
        self.results = self.tree.intersect(self.rays, nthreads=self.nthreads)
        lib().orc_spath_trace(_p(self.out_tex), _p(self.rays), _p(self.origin_rays), _p(self.results),
                              _p(self.source_results), _p(self.progress), _p(self.random), _p(s.idx4),
                              _p(s.normals4), _p(s.meshes), _p(s.materials), C.c_uint32(self.n),
                              C.c_uint32(rv), C.c_int(self.nthreads))

    def resolve(self) -> np.ndarray:
        out = np.zeros(self.n, np.uint32)
        lib().orc_resolve_avg(_p(out), _p(self.out_tex), C.c_uint32(self.n))
        return out


def cast_ray_test(scene, rays: np.ndarray, nthreads: int = 0) -> np.ndarray:
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
    out = np.zeros(rays.shape[0], np.uint32)
    lib().orc_cast_ray_test(_p(out), _p(rays), C.c_uint32(rays.shape[0]), _p(scene.idx4), _p(scene.verts4),
                            _p(scene.normals4), _p(scene.meshes), C.c_uint32(scene.meshes.shape[0]),
                            _p(scene.materials), _p(scene.spheres), C.c_uint32(scene.spheres.shape[0]),
                            _p(scene.planes), C.c_uint32(scene.planes.shape[0]), C.c_int(nthreads or max_threads()))
    return out


def path_trace(scene, rays: np.ndarray, w: int, h: int, cam_pos, magic: int, nthreads: int = 0) -> np.ndarray:
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
    out = np.zeros((w * h, 4), np.float32)
    cam = np.asarray(cam_pos, np.float32)
    lib().orc_path_trace(_p(out), _p(rays), C.c_uint32(w), C.c_uint32(h), _p(cam), C.c_uint32(magic),
                         _p(scene.idx4), _p(scene.verts4), _p(scene.normals4), _p(scene.meshes),
                         C.c_uint32(scene.meshes.shape[0]), _p(scene.materials), _p(scene.spheres),
                         C.c_uint32(scene.spheres.shape[0]), _p(scene.planes), C.c_uint32(scene.planes.shape[0]),
                         C.c_int(nthreads or max_threads()))
    return out


def f_buffer_average(out_tex: np.ndarray, fresh: np.ndarray, sample: int) -> None:
    lib().orc_f_buffer_average(_p(out_tex), _p(fresh), C.c_uint32(out_tex.shape[0]), C.c_uint32(sample))


def f_to_byte(tex: np.ndarray) -> np.ndarray:
    out = np.zeros(tex.shape[0], np.uint32)
    lib().orc_f_to_byte(_p(out), _p(tex), C.c_uint32(tex.shape[0]))
    return out


def kd_ray_draw(rays: np.ndarray) -> np.ndarray:
    """kdtree_ray_draw, src/kernels/kdtree.cl:323-341: (origin + 1) / 2 as a colour."""
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
    tex = np.ones((rays.shape[0], 4), np.float32)
    tex[:, :3] = (rays[:, 0:3] + np.float32(1)) / np.float32(2)
    return f_to_byte(tex)


def kd_test_draw(scene, results: np.ndarray) -> np.ndarray:
    """kdtree_test_draw, src/kernels/kdtree.cl:344-384: the normal interpolated at each hit, (n + 1) / 2 as a
    colour; a miss (t == 0) is get_colour(0)."""
    idx = scene.idx4.reshape(-1, 4)
    tri = results["tri"].astype(np.int64)
    u, v = results["u"][:, None], results["v"][:, None]
    n1, n2, n3 = (scene.normals4[idx[tri + k, 1], :3] for k in range(3))
    normal = (n1 * ((np.float32(1) - u) - v) + n2 * u) + n3 * v
    tex = np.ones((len(tri), 4), np.float32)
    tex[:, :3] = (normal + np.float32(1)) / np.float32(2)
    tex[results["t"] == 0] = 0
    return f_to_byte(tex)
