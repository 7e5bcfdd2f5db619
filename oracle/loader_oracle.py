"""Restatement of the reference's scene assembly (src/loader.c:140-450) in numpy.

TEST INFRASTRUCTURE ONLY: never imported by the product.  The OBJ files are parsed by the parser the reference
vendors (oracle/_ref/libref_tinyobj.so, built from the reference checkout), so this module only runs where that
library exists; what is restated here is what the reference's load_obj / load_scene_json do with the parser's
output: one mesh per shape, index tuples {v, vn, vt, mesh} offset by the elements of the files before it,
vertices and normals padded to four floats, min/max over the referenced vertices, model = T * (S * R).
"""
import json
import os

import numpy as np

from . import ref_obj_parse


def _m4_mul(a, b):
    """libs/vec.h:1240-1262: element (i, j) at 4 i + j, sum left to right in single precision."""
    a = a.astype(np.float32); b = b.astype(np.float32)
    o = np.zeros(16, np.float32)
    for i in range(4):
        for j in range(4):
            acc = np.float32(a[4 * i + 0] * b[0 + j])
            for k in (1, 2, 3):
                acc = np.float32(acc + np.float32(a[4 * i + k] * b[4 * k + j]))
            o[4 * i + j] = acc
    return o


def _identity():
    m = np.zeros(16, np.float32)
    m[[0, 5, 10, 15]] = 1
    return m


def _from_quat(q):
    """libs/vec.h:1162-1198 + 1563-1576"""
    f = np.float32
    qx, qy, qz, qw = (f(v) for v in q)
    x2, y2, z2 = f(qx + qx), f(qy + qy), f(qz + qz)
    xx, xy, xz = f(qx * x2), f(qx * y2), f(qx * z2)
    yy, yz, zz = f(qy * y2), f(qy * z2), f(qz * z2)
    wx, wy, wz = f(qw * x2), f(qw * y2), f(qw * z2)
    m = _identity()
    m[0], m[1], m[2] = f(1) - f(yy + zz), f(xy - wz), f(xz + wy)
    m[4], m[5], m[6] = f(xy + wz), f(1) - f(xx + zz), f(yz - wx)
    m[8], m[9], m[10] = f(xz - wy), f(yz + wx), f(1) - f(xx + yy)
    return m


def load_scene(url: str, root: str) -> dict:
    """dict of arrays in the layouts of raytracer_amd.scenes (verts4, normals4, idx4, meshes fields, ...)."""
    desc = json.load(open(os.path.join(root, url)))
    num = lambda o, k: float(o.get(k, 0)) if isinstance(o.get(k, 0), (int, float)) and not isinstance(o.get(k, 0), bool) else 0.0  # noqa: E731
    out = {}
    mats = desc.get("materials", [])
    out["materials"] = np.array([[num(m, "r"), num(m, "g"), num(m, "b"), num(m, "reflectivity")] for m in mats], np.float32).reshape(-1, 4)
    prims = desc.get("primitives", {})
    out["spheres"] = np.array([[num(s, "x"), num(s, "y"), num(s, "z"), num(s, "radius")] for s in prims.get("spheres", [])], np.float32).reshape(-1, 4)
    out["sphere_mat"] = np.array([int(num(s, "mat_index")) for s in prims.get("spheres", [])], np.int32)
    out["planes"] = np.array([[num(p, k) for k in ("x", "y", "z", "nx", "ny", "nz")] for p in prims.get("planes", [])], np.float32).reshape(-1, 6)
    out["plane_mat"] = np.array([int(num(p, "mat_index")) for p in prims.get("planes", [])], np.int32)

    verts, nrmls, texs, idx, meshes = [], [], [], [], []
    vert_off = nrml_off = tex_off = 0
    for m in desc.get("meshes", []):
        o = ref_obj_parse(open(os.path.join(root, m["url"]), "rb").read())
        tr, sc, rot = _identity(), _identity(), _identity()
        if "px" in m:
            tr[12], tr[13], tr[14] = num(m, "px"), num(m, "py"), num(m, "pz")
        if "sx" in m:
            sc[0], sc[5], sc[10] = num(m, "sx"), num(m, "sy"), num(m, "sz")
        if "rw" in m:
            rot = _from_quat([num(m, "rx"), num(m, "ry"), num(m, "rz"), num(m, "rw")])
        model = _m4_mul(tr, _m4_mul(sc, rot))
        faces = o["faces"]                                       # {v, vt, vn} per corner
        for off, length in zip(o["shape_face_offset"], o["shape_length"]):
            corners = faces[3 * int(off): 3 * (int(off) + int(length))]
            used = o["vertices"][corners[:, 0]]
            tup = np.stack([vert_off + corners[:, 0], nrml_off + corners[:, 2], tex_off + corners[:, 1],
                            np.full(len(corners), len(meshes))], axis=1).astype(np.int64)
            meshes.append({"model": model, "min": used.min(axis=0), "max": used.max(axis=0),
                           "index_offset": sum(len(t) for t in idx), "num_indices": 3 * int(length),
                           "material_index": int(num(m, "mat_index"))})
            idx.append((tup & 0xffffffff).astype(np.uint32).view(np.int32))
        verts.append(o["vertices"]); nrmls.append(o["normals"]); texs.append(o["texcoords"])
        vert_off += len(o["vertices"]); nrml_off += len(o["normals"]); tex_off += len(o["texcoords"])

    def pad4(parts):
        a = np.concatenate(parts) if parts else np.zeros((0, 3), np.float32)
        return np.concatenate([a, np.zeros((len(a), 1), np.float32)], axis=1)
    out["verts4"] = pad4(verts)
    out["normals4"] = pad4(nrmls)
    out["texcoords"] = np.concatenate(texs) if texs else np.zeros((0, 2), np.float32)
    out["idx4"] = np.concatenate(idx) if idx else np.zeros((0, 4), np.int32)
    out["meshes"] = meshes
    return out
