"""The traversal entry point that allocates nothing (rt_k_kd_intersect_noalloc: its work counter lives in a
file-scope device array of the library instead of allocated memory) against rt_k_kd_intersect, through the C ABI:
same hit records bit for bit, on two calls in a row, the second on fewer rays, so that what the first call left
in the shared array must not matter.  Collected last: the other GPU tests do not depend on it."""
import ctypes as C

import numpy as np
import pytest

from helpers import assert_bits_equal

pytestmark = pytest.mark.gpu


def _rays(n, seed):
    rng = np.random.RandomState(seed)
    o = rng.uniform(-12, 12, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[: n // 50, 1] = 0.0                      # some axis-parallel rays
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3], rays[:, 4:7] = o, d
    rays[n // 2: n // 2 + 7] = 0.0             # null rays, as the first render iteration traces
    return rays


def test_noalloc_traversal_equals_allocating_entry_point(rt, orc):
    sc = rt.scenes.icosphere_scene(4)
    tree = orc.KdTree(sc.verts4, sc.idx4)
    r = rt.Renderer(sc, 8, 8).build()
    r.L.raytracer_prepass(r.rctx)              # uploads the flattened tree
    kd = C.byref(r.kdt.view)
    n_max = 70001
    ro = r.L.rt_mem_alloc(r.dev, 16 * n_max); rd = r.L.rt_mem_alloc(r.dev, 16 * n_max)
    h_a = r.L.rt_mem_alloc(r.dev, 16 * n_max); h_b = r.L.rt_mem_alloc(r.dev, 16 * n_max)
    try:
        for n, seed in ((n_max, 11), (4099, 12)):            # the second call is the smaller one
            rays = _rays(n, seed)
            r.upload(ro, np.ascontiguousarray(rays[:, 0:4])); r.upload(rd, np.ascontiguousarray(rays[:, 4:8]))
            r.L.rt_mem_zero(r.dev, h_a, 16 * n_max); r.L.rt_mem_zero(r.dev, h_b, 16 * n_max)
            r.L.rt_k_kd_intersect(r.dev, kd, ro, rd, h_a, n)
            assert r.L.rt_k_kd_intersect_noalloc(r.dev, kd, ro, rd, h_b, n) == 0
            r.sync()
            want = r.download(h_a, np.uint32, 4 * n_max).reshape(n_max, 4)
            got = r.download(h_b, np.uint32, 4 * n_max).reshape(n_max, 4)
            assert_bits_equal(got, want, f"hit records, {n} rays")
            assert not got[n:].any()                          # nothing written past the rays of the call
            cpu = tree.intersect(rays)
            assert_bits_equal(got[:n, 0], cpu["tri"], "index offset vs oracle")
            assert_bits_equal(got[:n, 1], cpu["t"].view(np.uint32), "t vs oracle")
            assert 0.05 < (cpu["t"] != 0).mean() < 0.95
        # over the limit: refused with an error code, nothing launched
        assert r.L.rt_k_kd_intersect_noalloc(r.dev, kd, ro, rd, h_b, 0x80000001) == 1
        assert r.L.rt_k_kd_intersect_noalloc(r.dev, kd, ro, rd, None, 64) == 2
        assert r.L.rt_k_kd_intersect_noalloc(r.dev, kd, ro, rd, h_b, 0) == 0
        r.sync()
    finally:
        for m in (ro, rd, h_a, h_b):
            r.L.rt_mem_free(r.dev, m)
        r.close()


def test_two_devices_on_one_gpu_have_counters_of_their_own(rt, orc):
    """Two rt_devices opened on the same GPU (two Renderers) call the entry point at the same time: each has its
    own counter lines in the GPU's static array, so neither call clears the counter the other's waves claim rays
    from -- every ray of both calls gets its hit record (the calls are queued without a host synchronisation
    between them, on the two devices' own streams, several times over)."""
    sc = rt.scenes.icosphere_scene(4)
    a = rt.Renderer(sc, 8, 8).build(); a.L.raytracer_prepass(a.rctx)
    b = rt.Renderer(sc, 8, 8).build(); b.L.raytracer_prepass(b.rctx)
    n = 200_000
    rays = _rays(n, 21)
    bufs = []
    try:
        for r in (a, b):
            ro = r.L.rt_mem_alloc(r.dev, 16 * n); rd = r.L.rt_mem_alloc(r.dev, 16 * n)
            want = r.L.rt_mem_alloc(r.dev, 16 * n); got = r.L.rt_mem_alloc(r.dev, 16 * n)
            r.upload(ro, np.ascontiguousarray(rays[:, 0:4])); r.upload(rd, np.ascontiguousarray(rays[:, 4:8]))
            r.L.rt_k_kd_intersect(r.dev, C.byref(r.kdt.view), ro, rd, want, n)
            bufs.append((r, ro, rd, want, got))
        a.sync(); b.sync()
        for rep in range(6):
            for r, ro, rd, want, got in bufs:
                r.L.rt_mem_zero(r.dev, got, 16 * n)
            for r, ro, rd, want, got in bufs:            # queued back to back: the two launches overlap on the GPU
                assert r.L.rt_k_kd_intersect_noalloc(r.dev, C.byref(r.kdt.view), ro, rd, got, n) == 0
            for r, ro, rd, want, got in bufs:
                r.sync()
                assert_bits_equal(r.download(got, np.uint32, 4 * n), r.download(want, np.uint32, 4 * n),
                                  f"hit records, round {rep}")
    finally:
        for r, *mems in bufs:
            for m in mems:
                r.L.rt_mem_free(r.dev, m)
        a.close(); b.close()


def test_device_without_counter_lines_is_refused_not_shared(rt):
    """The static array holds counter lines for 16 rt_devices per GPU: a device opened while 16 others are open
    gets none and its calls return 3 (nothing launched); closing a device frees its lines for the next one."""
    L = rt.lib()
    sc = rt.scenes.icosphere_scene(2)
    r = rt.Renderer(sc, 8, 8).build(); r.L.raytracer_prepass(r.rctx)
    view = C.byref(r.kdt.view)                  # device memory of GPU 0: any rt_device on it can traverse the tree
    ro = L.rt_mem_alloc(r.dev, 1024); rd = L.rt_mem_alloc(r.dev, 1024); h = L.rt_mem_alloc(r.dev, 1024)
    L.rt_mem_zero(r.dev, ro, 1024); L.rt_mem_zero(r.dev, rd, 1024)
    r.sync()
    devs, refused = [], None
    try:
        assert L.rt_k_kd_intersect_noalloc(r.dev, view, ro, rd, h, 64) == 0
        for _ in range(16):                     # other tests may hold devices too: the 16th open at the latest is refused
            d = L.rt_device_open(0)
            devs.append(d)
            rc = L.rt_k_kd_intersect_noalloc(d, view, ro, rd, h, 64)
            L.rt_device_sync(d)
            if rc == 3:
                refused = d
                break
            assert rc == 0
        assert refused is not None
        assert L.rt_k_kd_intersect_noalloc(refused, view, ro, rd, None, 64) == 2      # argument checks come first
        L.rt_device_close(devs.pop(0))          # gives its lines back
        extra = L.rt_device_open(0)
        devs.append(extra)
        assert L.rt_k_kd_intersect_noalloc(extra, view, ro, rd, h, 64) == 0
        L.rt_device_sync(extra)
        assert L.rt_k_kd_intersect_noalloc(refused, view, ro, rd, h, 64) == 3        # decided when it was opened
    finally:
        for d in devs:
            L.rt_device_close(d)
        r.sync()
        for m in (ro, rd, h):
            L.rt_mem_free(r.dev, m)
        r.close()
