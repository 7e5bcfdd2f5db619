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


def load_ref_render_fixture(rt):
    """tests/golden/ref_render_path_obj2.npz: the reference's scene path_obj2 as loaded arrays, and the emitter /
    object masks of the render of it that the reference stores (renders/3_custom_2046.PNG, 1920x1080)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_render_path_obj2.npz"))
    S = rt.scenes
    sc = S.SceneArrays(verts4=z["verts4"], normals4=z["normals4"], idx4=z["idx4"],
                       meshes=z["meshes"].view(S.MESH_DT).reshape(-1), materials=z["materials"].view(S.MATERIAL_DT).reshape(-1),
                       spheres=np.zeros(0, S.SPHERE_DT), planes=np.zeros(0, S.PLANE_DT))
    w, h = int(z["width"]), int(z["height"])
    emitter = np.unpackbits(z["emitter_bits"])[:w * h].astype(bool)
    objects = np.unpackbits(z["object_bits"])[:w * h].astype(bool)
    return sc, w, h, emitter, objects


def primary_material_classes(sc, source_results):
    """Material index of the surface each primary ray hits (-1: nothing)."""
    mesh = sc.idx4.reshape(-1, 4)[source_results["tri"], 3]
    mat = np.asarray(sc.meshes["material_index"])[mesh]
    return np.where(source_results["t"] != 0, mat, -1)


def check_against_stored_render(sc, source_results, emitter, objects):
    """Primary visibility against the reference's stored frame: the emitter silhouette pixel for pixel; the
    orange objects wherever the stored frame shows their colour (their parts in deep shadow carry no colour to
    tell them by, so the stored mask is a subset)."""
    cls = primary_material_classes(sc, source_results)
    assert np.array_equal(cls == 1, emitter), f"{int(((cls == 1) != emitter).sum())} emitter pixels differ"
    mine = cls == 3
    assert not (objects & ~mine).any(), f"{int((objects & ~mine).sum())} object-coloured pixels hit no object"
    iou = (mine & objects).sum() / (mine | objects).sum()
    assert iou > 0.95, iou
