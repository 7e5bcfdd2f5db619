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
