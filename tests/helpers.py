"""Shared helpers for the parity tests."""
import numpy as np


def bits(a: np.ndarray) -> np.ndarray:
    """Bit pattern view so that float comparisons are exact (NaN == NaN, -0 != +0)."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        return a.view(np.uint32)
    return a


def assert_bits_equal(got, want, what=""):
    g, w = bits(got), bits(want)
    assert g.shape == w.shape, f"{what}: shape {g.shape} vs {w.shape}"
    bad = np.nonzero(g != w)
    n = len(bad[0])
    if n:
        first = tuple(int(b[0]) for b in bad)
        raise AssertionError(f"{what}: {n} of {g.size} words differ; first at {first}: "
                             f"got {got[first]!r} want {want[first]!r}")


def hits_equal(got, want, what="hits"):
    for f in ("tri", "t", "u", "v"):
        assert_bits_equal(got[f], want[f], f"{what}.{f}")


def oracle_spath_run(orc, scene, tree, w, h, seed, iterations, y0=0, full_h=None):
    """Drive the oracle through prepass + iterations with the counter-based RNG
    streams the library uses in RT_RNG_SPLITMIX mode."""
    from raytracer_amd.scenes import splitmix32
    st = orc.SpathState(scene, tree, w, h, y0, full_h)
    first = y0 * w
    rb = splitmix32(seed, np.arange(first, first + w * h, dtype=np.uint64))
    rv = splitmix32(seed + 1, np.arange(0, iterations + 1, dtype=np.uint64))
    st.prepass(rb, int(rv[0]))
    for k in range(iterations):
        st.iterate(int(rv[k + 1]))
    return st
