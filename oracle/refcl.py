"""TEST INFRASTRUCTURE: run the reference's own OpenCL C kernels on the MI355X.

oracle/build_ref_kernels.py compiles the reference's kernel files, unmodified,
into gfx950 code objects under oracle/_ref/ (they travel to the GPU box as
built files; the sources do not).  This module launches them through
oracle/hsa_runner.c and mirrors the reference's host call sequences -- the
argument lists of src/spath_raytracer.c:136-171,252-353, src/raytracer.c:182-206,
src/ss_raytracer.c and src/path_raytracer.c -- so that tests can compare what the
reference's kernels compute with the oracle restatement and with the product.

What stands in for src/parallel.c's image objects: the OpenCL runtime of this
platform has no image support on gfx950, but a 1-D image-buffer read in the
compiled kernel is a typed buffer load through a 4-dword buffer descriptor
fetched from the image argument's address.  `image1d` builds that descriptor
for a CL_RGBA / CL_FLOAT image buffer (gen_1d_image_buffer, src/parallel.c:282-324):
16-byte texels, 32_32_32_32 FLOAT, one record per texel, reads past the end
return zero.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

HIT_DT = np.dtype([("tri", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])


class _Kernel(C.Structure):
    _fields_ = [("object", C.c_uint64), ("kernarg_size", C.c_uint32), ("group_size", C.c_uint32),
                ("private_size", C.c_uint32), ("dynamic_stack", C.c_uint32)]


def available(tag: str = "ico4", flavour: str = "strict") -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", f"refcl_{tag}_{flavour}.hsaco"))


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libhsa_runner.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "libhsa_runner.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(path)
        L.hr_last_error.restype = C.c_char_p
        L.hr_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
        L.hr_free.argtypes = [C.c_void_p]
        L.hr_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.hr_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.hr_fill32.argtypes = [C.c_void_p, C.c_uint32, C.c_size_t]
        L.hr_load.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.hr_kernel_info.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(_Kernel)]
        L.hr_dispatch.argtypes = [C.POINTER(_Kernel), C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_uint32), C.c_double, C.POINTER(C.c_double)]
        _LIB = L
    return _LIB


class LauncherError(RuntimeError):
    """A call of the HSA launcher failed; `what` names the call, `rc` is its return code (hr_dispatch: -9 means
    the kernel was still running when the wait gave up -- its buffers must not be released any more)."""

    def __init__(self, what, rc, message):
        super().__init__(f"{what}: {message} (rc {rc})")
        self.what, self.rc = what, rc

    @property
    def setup(self):
        """True when the launcher could not be set up at all (no HSA agent, code object rejected): nothing ran."""
        return self.what in ("hr_init", "hr_load")


def _ok(rc, what):
    if rc != 0:
        raise LauncherError(what, rc, _lib().hr_last_error().decode())


class DevBuf:
    """Device memory of the launcher (freed with the object)."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _ok(_lib().hr_alloc(max(self.nbytes, 4), C.byref(p)), "hr_alloc")
        self.ptr = int(p.value)

    @classmethod
    def of(cls, a: np.ndarray) -> "DevBuf":
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        b.write(a)
        return b

    @classmethod
    def zeros(cls, nbytes: int) -> "DevBuf":
        b = cls((nbytes + 3) // 4 * 4)
        b.zero()
        return b

    def write(self, a: np.ndarray) -> None:
        a = np.ascontiguousarray(a)
        assert a.nbytes <= max(self.nbytes, 4)
        if a.nbytes:
            _ok(_lib().hr_h2d(self.ptr, a.ctypes.data, a.nbytes), "hr_h2d")

    def zero(self) -> None:
        _ok(_lib().hr_fill32(self.ptr, 0, max(self.nbytes, 4) // 4), "hr_fill32")

    def read(self, dtype, count: int) -> np.ndarray:
        out = np.zeros(count, dtype)
        assert out.nbytes <= max(self.nbytes, 4)
        if out.nbytes:
            _ok(_lib().hr_d2h(out.ctypes.data, self.ptr, out.nbytes), "hr_d2h")
        return out

    def __del__(self):
        try:
            if self.ptr:
                _lib().hr_free(self.ptr)
                self.ptr = 0
        except Exception:
            pass


def image1d(data: DevBuf, texels: int) -> DevBuf:
    """Buffer descriptor of a CL_RGBA/CL_FLOAT 1-D image buffer over `data` (gfx9 layout):
    word0/1 base address, stride 16 in word1[29:16], word2 record count, word3 dst_sel XYZW |
    NUM_FORMAT FLOAT (7 << 12) | DATA_FORMAT 32_32_32_32 (14 << 15)."""
    assert texels * 16 <= max(data.nbytes, 4) and data.ptr % 16 == 0 and data.ptr < (1 << 48)
    words = np.array([data.ptr & 0xffffffff, ((data.ptr >> 32) & 0xffff) | (16 << 16), texels,
                      (4 | (5 << 3) | (6 << 6) | (7 << 9)) | (7 << 12) | (14 << 15), 0, 0, 0, 0], np.uint32)
    d = DevBuf.of(words)
    d.keep = data
    return d


class RefProgram:
    """One compiled flavour of the reference's device program for one scene."""

    def __init__(self, tag: str, flavour: str = "strict"):
        base = os.path.join(_HERE, "_ref", f"refcl_{tag}_{flavour}")
        self.meta = json.load(open(base + ".json"))
        image = open(base + ".hsaco", "rb").read()
        L = _lib()
        _ok(L.hr_init(), "hr_init")
        m = C.c_void_p()
        _ok(L.hr_load(image, len(image), C.byref(m)), "hr_load")
        self._module = m
        self._kernels = {}
        self.simt = int(self.meta["simt_width"])
        self.last_ms = {}

    def _kernel(self, name):
        if name not in self._kernels:
            k = _Kernel()
            _ok(_lib().hr_kernel_info(self._module, self.meta["kernels"][name]["symbol"].encode(), C.byref(k)),
                f"hr_kernel_info({name})")
            md = self.meta["kernels"][name]
            assert k.kernarg_size == md["kernarg_size"] and k.private_size >= md["private_size"]
            self._kernels[name] = k
        return self._kernels[name]

    def launch(self, name: str, args: list, global_size: int, local_size: int = 64, timeout_ms: float = 30000.0):
        """args: one entry per explicit kernel argument, in order: DevBuf (pointer / image), int (32-bit
        by-value) or bytes (by-value of the argument's size).  global_size / local_size: an int or a tuple."""
        md = self.meta["kernels"][name]
        k = self._kernel(name)
        block = bytearray(md["kernarg_size"])
        explicit = [a for a in md["args"] if not a["kind"].startswith("hidden")]
        if len(explicit) != len(args):
            raise ValueError(f"{name}: {len(explicit)} arguments expected, {len(args)} given")
        for a, v in zip(explicit, args):
            if a["kind"] in ("global_buffer", "image"):
                if not isinstance(v, DevBuf):
                    raise TypeError(f"{name}: argument at {a['offset']} needs device memory")
                raw = struct.pack("<Q", v.ptr)
            elif isinstance(v, (bytes, bytearray)):
                raw = bytes(v)
            else:
                raw = struct.pack("<I", int(v) & 0xffffffff)
            if len(raw) != a["size"]:
                raise ValueError(f"{name}: argument at {a['offset']} is {a['size']} bytes, got {len(raw)}")
            block[a["offset"]:a["offset"] + a["size"]] = raw
        gs = tuple(global_size) if isinstance(global_size, (tuple, list)) else (int(global_size),)
        ls = tuple(local_size) if isinstance(local_size, (tuple, list)) else (int(local_size),)
        gs, ls = gs + (1,) * (3 - len(gs)), ls + (1,) * (3 - len(ls))
        if any(g % l for g, l in zip(gs, ls)):
            raise ValueError("global size must be a multiple of the local size")
        hidden = {"hidden_grid_dims": 2 if gs[1] > 1 else 1}
        for i, ax in enumerate("xyz"):
            hidden[f"hidden_block_count_{ax}"] = gs[i] // ls[i]
            hidden[f"hidden_group_size_{ax}"] = ls[i]
        for a in md["args"]:
            if a["kind"] in hidden:
                block[a["offset"]:a["offset"] + a["size"]] = int(hidden[a["kind"]]).to_bytes(a["size"], "little")
        grid = (C.c_uint32 * 3)(*gs)
        wg = (C.c_uint32 * 3)(*ls)
        ms = C.c_double(0)
        buf = (C.c_char * len(block)).from_buffer(block)
        _ok(_lib().hr_dispatch(C.byref(k), buf, len(block), grid, wg, float(timeout_ms), C.byref(ms)), f"dispatch {name}")
        self.last_ms[name] = ms.value
        return ms.value


_IDENTITY_MASK = np.array([0xffffffff, 0, 0, 0, 0, 0xffffffff, 0, 0, 0, 0, 0xffffffff, 0, 0, 0, 0, 0xffffffff, 0], np.uint32)


class RefScene:
    """Scene resources as scene_init_resources / scene_resource_push upload them (src/scene.c:7-145)
    plus the serialised kd-tree; `blob` is the wire format (oracle's or the product's, they are equal)."""

    def __init__(self, prog: RefProgram, sc, blob: np.ndarray | None):
        self.prog, self.sc = prog, sc
        self.materials = DevBuf.of(sc.materials)
        self.meshes = DevBuf.of(sc.meshes)
        self.spheres = DevBuf.of(sc.spheres)
        self.planes = DevBuf.of(sc.planes)
        self._idx = DevBuf.of(sc.idx4)
        self._verts = DevBuf.of(sc.verts4)
        self._normals = DevBuf.of(sc.normals4)
        self.indices = image1d(self._idx, sc.idx4.shape[0])
        self.vertices = image1d(self._verts, sc.verts4.shape[0])
        self.normals = image1d(self._normals, sc.normals4.shape[0])
        self.kd_tree = DevBuf.of(blob) if blob is not None else None
        self.dumb_data = DevBuf.of(_IDENTITY_MASK)

    def generate_rays(self, rays: DevBuf, w: int, h: int) -> None:
        """_raytracer_gen_ray_buffer, src/raytracer.c:182-206 (the camera matrix is passed and ignored)."""
        self.prog.launch("generate_rays", [rays, w, h, np.eye(4, dtype=np.float32).tobytes()], w * h)

    def kd_intersect(self, out: DevBuf, rays: DevBuf, n: int, groups: int = 2048) -> float:
        """bad_buf_update + spath_raytracer_kd_collision, src/spath_raytracer.c:15-37,136-171: persistent
        one-warp workgroups; the reference launches num_cores*16 work items."""
        self.dumb_data.write(_IDENTITY_MASK)
        g = min(groups, (n + self.prog.simt - 1) // self.prog.simt) * self.prog.simt
        return self.prog.launch("kdtree_intersection", [out, rays, self.dumb_data, self.meshes, self.indices,
                                                         self.vertices, self.kd_tree, n], g, self.prog.simt)

    def cast_ray_test(self, out: DevBuf, rays: DevBuf, w: int, h: int, pos=(0.0, 0.0, 0.0, 0.0)) -> float:
        """ss_raytracer_render, src/ss_raytracer.c:7-56."""
        return self.prog.launch("cast_ray_test", [out, rays, self.materials, self.spheres, self.planes, self.meshes,
                                                   self.indices, self.vertices, self.normals, w, h,
                                                   np.asarray(pos, np.float32).tobytes()], w * h)


    def path_trace(self, fresh: DevBuf, rays: DevBuf, w: int, h: int, pos, magic: int) -> float:
        """path_raytracer_trace, src/path_raytracer.c:28-80: a two-dimensional launch over the frame."""
        return self.prog.launch("path_trace", [fresh, rays, self.materials, self.spheres, self.planes, self.meshes,
                                                self.indices, self.vertices, self.normals, w,
                                                np.asarray(list(pos) + [0.0], np.float32)[:4].tobytes(), magic],
                                (w, h), (8, 8))

    def f_buffer_average(self, out_tex: DevBuf, fresh: DevBuf, w: int, h: int, num_samples: int, sample: int) -> None:
        """path_raytracer_average_buffers, src/path_raytracer.c:82-107"""
        self.prog.launch("f_buffer_average", [out_tex, fresh, w, h, num_samples, sample], w * h)

    def f_buffer_to_byte_buffer(self, out: DevBuf, tex: DevBuf, w: int, h: int) -> None:
        """path_raytracer_push_path, src/path_raytracer.c:109-140"""
        self.prog.launch("f_buffer_to_byte_buffer", [out, tex, w, h], w * h)


class RefSpath:
    """The split path tracer's buffers and call order (src/spath_raytracer.c:39-116,355-436) on the
    reference's kernels; same interface as oracle.SpathState."""

    def __init__(self, scene: RefScene, w: int, h: int):
        assert (w * h) % 64 == 0
        self.s, self.w, self.h, self.n = scene, w, h, w * h
        n = self.n
        self.d_out_tex = DevBuf.zeros(16 * n)
        self.d_rays = DevBuf.zeros(32 * n)
        self.d_origin_rays = DevBuf.zeros(32 * n)
        self.d_results = DevBuf.zeros(16 * n)
        self.d_source_results = DevBuf.zeros(16 * n)
        self.d_progress = DevBuf.zeros(48 * n)
        self.d_random = DevBuf.zeros(4 * n)
        self.d_output = DevBuf.zeros(4 * n)
        self.iteration = 0
        self.kd_ms = []
        self.trace_ms = []

    # host-side views
    rays = property(lambda self: self.d_rays.read(np.float32, 8 * self.n).reshape(self.n, 8))
    origin_rays = property(lambda self: self.d_origin_rays.read(np.float32, 8 * self.n).reshape(self.n, 8))
    results = property(lambda self: self.d_results.read(HIT_DT, self.n))
    source_results = property(lambda self: self.d_source_results.read(HIT_DT, self.n))
    progress = property(lambda self: self.d_progress.read(np.uint32, 12 * self.n).reshape(self.n, 12))
    out_tex = property(lambda self: self.d_out_tex.read(np.float32, 4 * self.n).reshape(self.n, 4))
    random = property(lambda self: self.d_random.read(np.uint32, self.n))

    def load_state(self, st) -> None:
        """Take over the per-pixel buffers of an oracle.SpathState (same record layouts)."""
        self.d_out_tex.write(st.out_tex)
        self.d_rays.write(st.rays)
        self.d_origin_rays.write(st.origin_rays)
        self.d_results.write(st.results)
        self.d_source_results.write(st.source_results)
        self.d_progress.write(st.progress)
        self.d_random.write(st.random)
        self.iteration = st.iteration

    def prepass(self, random_buffer: np.ndarray, rv_init: int, zero_rays: bool = True):
        s, p = self.s, self.s.prog
        self.d_out_tex.zero()
        self.iteration = 0
        self.d_progress.zero()
        s.generate_rays(self.d_rays, self.w, self.h)
        self.primary_rays = self.rays
        s.kd_intersect(self.d_results, self.d_rays, self.n)
        p.launch("segmented_path_trace_init",
                 [self.d_out_tex, self.d_rays, self.d_origin_rays, self.d_results, self.d_source_results,
                  self.d_progress, s.materials, s.meshes, s.indices, s.vertices, s.normals, self.w, rv_init], self.n)
        self.d_random.write(np.ascontiguousarray(random_buffer, np.uint32))
        self.first_bounce_rays = self.rays
        if zero_rays:                      # src/spath_raytracer.c:432-433
            self.d_rays.zero()

    def iterate(self, rv: int):
        s, p = self.s, self.s.prog
        p.launch("xorshift_batch", [self.d_random], self.n)
        self.iteration += 1
        self.kd_ms.append(s.kd_intersect(self.d_results, self.d_rays, self.n))
        self.trace_ms.append(p.launch(
            "segmented_path_trace",
            [self.d_out_tex, self.d_rays, self.d_origin_rays, self.d_results, self.d_source_results, self.d_progress,
             self.d_random, s.materials, s.meshes, s.indices, s.vertices, s.normals, self.w, rv], self.n))

    def resolve(self) -> np.ndarray:
        """spath_raytracer_avg_to_out, src/spath_raytracer.c:252-277"""
        self.s.prog.launch("f_buffer_to_byte_buffer_avg", [self.d_output, self.d_out_tex, self.d_progress, self.w, 0], self.n)
        return self.d_output.read(np.uint32, self.n)
