"""Synthetic scenes in the reference's host-side array layouts.

Every generator is deterministic (fixed seeds, no wall clock) and returns a
``SceneArrays`` whose members have exactly the byte layout the reference's
loader produces (``src/loader.c:140-226`` for the index tuple
``{v, vn, vt, mesh}`` and the 4-float padded vectors; ``include/scene.h:7-29``
and ``include/geom.h:49-75`` for material / mesh / sphere / plane records), so
they can be handed to the C API unchanged.

The reference's own large meshes are not redistributable test data (and the
biggest ones are missing from its checkout), so the bench and the parity tests
use these procedural stand-ins of the same shape: a displaced icosphere inside
a floor + back wall with an emissive quad above it, the layout of
``res/scenes/path_obj3.rsc``.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

MATERIAL_DT = np.dtype({"names": ["colour", "reflectivity"], "formats": [("<f4", 4), "<f4"],
                        "offsets": [0, 16], "itemsize": 32})
MESH_DT = np.dtype({"names": ["model", "max", "min", "index_offset", "num_indices", "material_index"],
                    "formats": [("<f4", 16), ("<f4", 4), ("<f4", 4), "<i4", "<i4", "<i4"],
                    "offsets": [0, 64, 80, 96, 100, 104], "itemsize": 128})
SPHERE_DT = np.dtype({"names": ["pos", "radius", "material_index"], "formats": [("<f4", 4), "<f4", "<i4"],
                      "offsets": [0, 16, 20], "itemsize": 32})
PLANE_DT = np.dtype({"names": ["pos", "norm", "material_index"], "formats": [("<f4", 4), ("<f4", 4), "<i4"],
                     "offsets": [0, 16, 32], "itemsize": 48})


@dataclass
class SceneArrays:
    verts4: np.ndarray      # float32 [nv, 4]
    normals4: np.ndarray    # float32 [nn, 4]
    idx4: np.ndarray        # int32   [ni, 4]  {v, vn, vt, mesh}
    meshes: np.ndarray      # MESH_DT [nm]
    materials: np.ndarray   # MATERIAL_DT
    spheres: np.ndarray     # SPHERE_DT
    planes: np.ndarray      # PLANE_DT

    @property
    def num_tris(self) -> int:
        return self.idx4.shape[0] // 3


# Materials of res/scenes/path_obj3.rsc:11-43 (index 1 has reflectivity 1,
# which the split path tracer treats as an emitter).
_PATH_MATERIALS = [
    (0.7, 0.7, 0.75, 0.7),
    (0.7, 0.7, 0.75, 1.0),
    (0.73, 0.9411, 0.819, 0.05),
    (0.8588, 0.352, 0.2588, 0.2),
    (1.0, 1.0, 1.0, 0.1),
]


def _materials(rows) -> np.ndarray:
    m = np.zeros(len(rows), MATERIAL_DT)
    for i, (r, g, b, refl) in enumerate(rows):
        m["colour"][i, :3] = (r, g, b)
        m["reflectivity"][i] = refl
    return m


def _pad4(a: np.ndarray) -> np.ndarray:
    out = np.zeros((a.shape[0], 4), np.float32)
    out[:, :3] = a
    return out


def _assemble(parts, materials, spheres=None, planes=None) -> SceneArrays:
    """parts: list of (verts[n,3], normals[m,3], faces_v[f,3], faces_n[f,3], material_index)."""
    verts, norms, idx, meshes = [], [], [], np.zeros(len(parts), MESH_DT)
    voff = noff = ioff = 0
    for mi, (v, n, fv, fn, mat) in enumerate(parts):
        v = np.asarray(v, np.float32)
        n = np.asarray(n, np.float32)
        fv = np.asarray(fv, np.int64)
        fn = np.asarray(fn, np.int64)
        rec = np.zeros((fv.shape[0] * 3, 4), np.int32)
        rec[:, 0] = (fv + voff).reshape(-1)
        rec[:, 1] = (fn + noff).reshape(-1)
        rec[:, 3] = mi
        used = v[fv.reshape(-1)]
        meshes["model"][mi] = np.eye(4, dtype=np.float32).reshape(-1)
        meshes["min"][mi, :3] = used.min(axis=0)
        meshes["max"][mi, :3] = used.max(axis=0)
        meshes["index_offset"][mi] = ioff
        meshes["num_indices"][mi] = rec.shape[0]
        meshes["material_index"][mi] = mat
        verts.append(_pad4(v)); norms.append(_pad4(n)); idx.append(rec)
        voff += v.shape[0]; noff += n.shape[0]; ioff += rec.shape[0]
    if parts:
        verts4 = np.ascontiguousarray(np.concatenate(verts))
        normals4 = np.ascontiguousarray(np.concatenate(norms))
        idx4 = np.ascontiguousarray(np.concatenate(idx))
    else:
        verts4 = np.zeros((0, 4), np.float32); normals4 = np.zeros((0, 4), np.float32)
        idx4 = np.zeros((0, 4), np.int32)
    return SceneArrays(verts4, normals4, idx4, meshes, materials,
                       spheres if spheres is not None else np.zeros(0, SPHERE_DT),
                       planes if planes is not None else np.zeros(0, PLANE_DT))


# --------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------
def icosphere(level: int):
    """Unit icosphere: (verts[n,3] float64, faces[f,3] int64), outward CCW winding."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t],
                  [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4],
                  [11, 10, 2], [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8],
                  [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], np.int64)
    for _ in range(level):
        nv = v.shape[0]
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        key = np.minimum(e[:, 0], e[:, 1]) * nv + np.maximum(e[:, 0], e[:, 1])
        uniq, inv = np.unique(key, return_inverse=True)
        a, b = uniq // nv, uniq % nv
        mid = v[a] + v[b]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        v = np.concatenate([v, mid])
        nf = f.shape[0]
        m01, m12, m20 = nv + inv[:nf], nv + inv[nf:2 * nf], nv + inv[2 * nf:]
        f = np.concatenate([np.stack([f[:, 0], m01, m20], 1), np.stack([f[:, 1], m12, m01], 1),
                            np.stack([f[:, 2], m20, m12], 1), np.stack([m01, m12, m20], 1)])
    return v, f


def _vertex_normals(v: np.ndarray, f: np.ndarray) -> np.ndarray:
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    n = np.zeros_like(v)
    for c in range(3):
        np.add.at(n, f[:, c], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    ln[ln == 0] = 1
    return n / ln


def _quad(p0, p1, p2, p3, normal):
    """Two triangles p0-p1-p2, p0-p2-p3 with one shared normal."""
    v = np.array([p0, p1, p2, p3], np.float32)
    n = np.array([normal], np.float32)
    fv = np.array([[0, 1, 2], [0, 2, 3]])
    fn = np.zeros((2, 3), np.int64)
    return v, n, fv, fn


def _room_parts():
    # emissive quad above and in front of the object, facing down/back, and a
    # floor + back wall: the roles of res/mesh/light.obj and
    # res/mesh/indigo_happy_walls.obj in res/scenes/path_obj3.rsc:45-83
    s = 1.0 / 2 ** 0.5
    light = _quad((-6, 7.0, -2.0), (6, 7.0, -2.0), (6, 3.0, 2.0), (-6, 3.0, 2.0), (0, -s, -s))
    floor = _quad((-30, -4.0, 20), (30, -4.0, 20), (30, -4.0, -40), (-30, -4.0, -40), (0, 1, 0))
    wall = _quad((-30, -16, -14.5), (30, -16, -14.5), (30, 16, -14.5), (-30, 16, -14.5), (0, 0, 1))
    parts = []
    for (v, n, fv, fn), mat in ((light, 1), (floor, 0), (wall, 0)):
        g = np.cross(v[fv[0, 1]] - v[fv[0, 0]], v[fv[0, 2]] - v[fv[0, 0]])
        if float(np.dot(g, n[0])) < 0:
            fv = fv[:, ::-1].copy()
        parts.append((v, n, fv, fn, mat))
    return parts


def icosphere_scene(level: int = 6, seed: int = 1) -> SceneArrays:
    """Displaced icosphere (20*4**level triangles) in the room of path_obj3.rsc.

    level 6 -> 81,920 triangles (the ~70k-triangle class), level 8 -> 1,310,720.
    """
    v, f = icosphere(level)
    rng = np.random.RandomState(seed)
    ph = rng.uniform(0, 2 * np.pi, size=(4, 3))
    fr = np.array([[3.0, 5.0, 4.0], [7.0, 6.0, 9.0], [13.0, 11.0, 12.0], [23.0, 19.0, 21.0]])
    amp = np.array([0.10, 0.05, 0.025, 0.0125])
    disp = np.zeros(v.shape[0])
    for a, w, p in zip(amp, fr, ph):
        disp += a * np.sin(v[:, 0] * w[0] + p[0]) * np.sin(v[:, 1] * w[1] + p[1]) * np.sin(v[:, 2] * w[2] + p[2])
    pos = v * (2.0 * (1.0 + disp))[:, None] + np.array([0.0, -1.0, -6.5])
    nrm = _vertex_normals(pos, f)
    parts = [(pos, nrm, f, f, 3)] + _room_parts()
    return _assemble(parts, _materials(_PATH_MATERIALS))


def soup_scene(num_tris: int, seed: int = 12345) -> SceneArrays:
    """Random triangle soup: centres uniform in a cube of half-width 1 around
    (0,0,-3), vertex offsets uniform +-0.3/cbrt(T); unshared vertices; all
    diffuse except an emissive ceiling quad."""
    rng = np.random.RandomState(seed)
    c = rng.uniform(-1, 1, size=(num_tris, 1, 3))
    o = rng.uniform(-1, 1, size=(num_tris, 3, 3)) * (0.3 / num_tris ** (1.0 / 3.0)) * 3.0
    v = (c + o).reshape(-1, 3) + np.array([0.0, 0.0, -3.0])
    f = np.arange(num_tris * 3, dtype=np.int64).reshape(-1, 3)
    g = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    ln = np.linalg.norm(g, axis=1, keepdims=True)
    ln[ln == 0] = 1
    n = g / ln
    fn = np.repeat(np.arange(num_tris, dtype=np.int64)[:, None], 3, axis=1)
    light = _quad((-3, 2.5, -6), (3, 2.5, -6), (3, 2.5, 0), (-3, 2.5, 0), (0, -1, 0))
    lv, ln_, lfv, lfn = light
    gl = np.cross(lv[lfv[0, 1]] - lv[lfv[0, 0]], lv[lfv[0, 2]] - lv[lfv[0, 0]])
    if float(np.dot(gl, ln_[0])) < 0:
        lfv = lfv[:, ::-1].copy()
    parts = [(v, n, f, fn, 3), (lv, ln_, lfv, lfn, 1)]
    return _assemble(parts, _materials(_PATH_MATERIALS))


def sphere_plane_scene() -> SceneArrays:
    """The primitives-only scene of res/scenes/path_test.rsc (3 spheres, 2
    planes, 5 materials, no meshes), typed in as data."""
    spheres = np.zeros(3, SPHERE_DT)
    for i, (x, y, z, r, m) in enumerate([(-0.4, 7, -2, 4, 1), (0, -0.6, -2.7, 0.5, 3), (-2, 0.2, -4, 1.7, 3)]):
        spheres["pos"][i, :3] = (x, y, z); spheres["radius"][i] = r; spheres["material_index"][i] = m
    planes = np.zeros(2, PLANE_DT)
    for i, (p, n, m) in enumerate([((0, -1, 0), (0, 1, 0), 2), ((0, 0, -10), (0, 0, 1), 2)]):
        planes["pos"][i, :3] = p; planes["norm"][i, :3] = n; planes["material_index"][i] = m
    return _assemble([], _materials(_PATH_MATERIALS), spheres, planes)


def mixed_scene(level: int = 2) -> SceneArrays:
    """Small mesh + the spheres and planes of path_test.rsc: exercises every
    branch of the brute-force intersector used by the SS and PATH integrators."""
    base = sphere_plane_scene()
    v, f = icosphere(level)
    pos = v * 0.8 + np.array([1.6, 0.1, -3.5])
    nrm = _vertex_normals(pos, f)
    return _assemble([(pos, nrm, f, f, 4)], _materials(_PATH_MATERIALS), base.spheres, base.planes)


def bundle_scene(copies=(45, 33)) -> SceneArrays:
    """Stacks of coincident triangles in the room of icosphere_scene: the builder cannot separate triangles that
    cover one another exactly, so each stack ends up in one leaf, longer than a leaf descriptor's count field holds
    (rt_hip.h: 31 = "the leaf begins with a header slot").  Of a stack a ray reports the first triangle."""
    shapes = [((-2.5, -2.0, -7.0), (0.5, -1.5, -6.0), (-1.0, 2.0, -8.0)),
              ((1.2, -2.5, -5.0), (3.6, -1.0, -6.5), (2.4, 1.5, -5.5)),
              ((-1.0, -3.0, -4.0), (1.0, -3.2, -4.5), (0.0, -1.0, -4.2))]
    parts = []
    for k, c in enumerate(copies):
        tri = np.array(shapes[k % len(shapes)], np.float32)
        g = np.cross(tri[1] - tri[0], tri[2] - tri[0])
        if g[2] < 0:                                    # face the camera (which looks down -z)
            tri = tri[::-1].copy()
            g = -g
        v = np.tile(tri, (c, 1))
        n = (g / np.linalg.norm(g))[None, :].astype(np.float32)
        fv = np.arange(3 * c, dtype=np.int64).reshape(-1, 3)
        parts.append((v, n, fv, np.zeros((c, 3), np.int64), 3 if k % 2 == 0 else 0))
    return _assemble(parts + _room_parts(), _materials(_PATH_MATERIALS))


def scene_from_npz(path: str) -> SceneArrays:
    """Scene arrays stored by the fixture scripts under tests/golden (np.savez of a loaded scene)."""
    z = np.load(path)

    def rec(name, dt):
        return z[name].view(dt).reshape(-1) if name in z.files and z[name].size else np.zeros(0, dt)
    return SceneArrays(verts4=z["verts4"], normals4=z["normals4"], idx4=z["idx4"], meshes=rec("meshes", MESH_DT),
                       materials=rec("materials", MATERIAL_DT), spheres=rec("spheres", SPHERE_DT),
                       planes=rec("planes", PLANE_DT))


def splitmix32(seed: int, index: np.ndarray) -> np.ndarray:
    """Counter-based 32-bit generator used to inject the per-pixel random
    buffer and the per-iteration random value deterministically (the reference
    draws both from libc rand() seeded by the wall clock,
    src/spath_raytracer.c:118-132,283,321,409)."""
    with np.errstate(over="ignore"):
        z = (np.asarray(index, np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
        z = (z + np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
        z = ((z ^ (z >> np.uint64(16))) * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
        z = ((z ^ (z >> np.uint64(13))) * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
        z = z ^ (z >> np.uint64(16))
    return z.astype(np.uint32)


def write_rsc(sc: SceneArrays, directory: str, name: str = "scene") -> str:
    """Writes `sc` as <directory>/<name>.rsc plus one Wavefront OBJ per mesh
    under <directory>/mesh/ (the file formats of the reference's res/ tree,
    src/loader.c:228-450) and returns the scene file's name relative to
    `directory`.  Coordinates are printed with 9 significant digits; the
    loader's own decimal conversion decides the floats that come back."""
    import json
    import os
    os.makedirs(os.path.join(directory, "mesh"), exist_ok=True)
    meshes = []
    for mi, m in enumerate(sc.meshes):
        first, count = int(m["index_offset"]), int(m["num_indices"])
        idx = sc.idx4[first:first + count]
        vs, inv_v = np.unique(idx[:, 0], return_inverse=True)
        ns, inv_n = np.unique(idx[:, 1], return_inverse=True)
        lines = [f"o mesh{mi}"]
        lines += ["v %.9g %.9g %.9g" % tuple(sc.verts4[v, :3]) for v in vs]
        lines += ["vn %.9g %.9g %.9g" % tuple(sc.normals4[n, :3]) for n in ns]
        for t in range(count // 3):
            c = [f"{inv_v[3 * t + j] + 1}//{inv_n[3 * t + j] + 1}" for j in range(3)]
            lines.append("f " + " ".join(c))
        url = f"mesh/{name}_{mi}.obj"
        with open(os.path.join(directory, url), "w") as f:
            f.write("\n".join(lines) + "\n")
        meshes.append({"url": url, "mat_index": int(m["material_index"])})
    desc = {
        "version": {"major": 0, "minor": 1, "type": "DEV"}, "name": name,
        "materials": [{"r": float(c[0]), "g": float(c[1]), "b": float(c[2]), "reflectivity": float(r)}
                      for c, r in zip(sc.materials["colour"], sc.materials["reflectivity"])],
        "meshes": meshes,
        "primitives": {
            "spheres": [{"x": float(s["pos"][0]), "y": float(s["pos"][1]), "z": float(s["pos"][2]),
                         "radius": float(s["radius"]), "mat_index": int(s["material_index"])} for s in sc.spheres],
            "planes": [{"x": float(p["pos"][0]), "y": float(p["pos"][1]), "z": float(p["pos"][2]),
                        "nx": float(p["norm"][0]), "ny": float(p["norm"][1]), "nz": float(p["norm"][2]),
                        "mat_index": int(p["material_index"])} for p in sc.planes]},
    }
    with open(os.path.join(directory, name + ".rsc"), "w") as f:
        json.dump(desc, f, indent=1)
    return name + ".rsc"
