#!/usr/bin/env python3
"""Writes the golden fixtures under tests/golden/.

kd_ico2.npz   serialised kd-tree of scenes.icosphere_scene(2) from the
              REFERENCE's own builder (oracle/_ref, built from /root/reference),
              plus the mask of bytes that carry data (the reference leaves
              struct padding uninitialised).  Needs the reference checkout.
spath_ico3.npz  oracle outputs for a 48x32 render of icosphere_scene(3):
              primary hits, and out_tex / progress after 6 iterations with the
              counter-based RNG streams (seed 11), in the oracle's default
              arithmetic (dot/cross/min/max as ROCm's OpenCL library defines
              them).  These pin the oracle against its own earlier output (the
              reference has no vectors for the kernels; its compiled kernels are
              held against the oracle on the GPU, tests/test_gpu_refkernels.py),
              so a change in them is a deliberate, reviewed event.
ss_sphere_plane.npz  oracle outputs for the sphere + plane scene of res/scenes/path_test.rsc at 96x64: the
              SS integrator's packed image and the PATH integrator's first sample (regression only,
              like spath_ico3.npz).
refkernel_ico4.npz  NOT written here: inputs and outputs of the reference's own kernels run on an MI355X,
              see make_refkernel_fixture.py.
obj_cases.npz  small hand-written OBJ texts (quads and n-gons, negative and missing
              indices, CRLF and lone-CR line endings, a last line without line ending,
              groups before / between / after faces, signs and positive exponents)
              and what the parser the REFERENCE vendors returns for each
              (oracle/_ref/libref_tinyobj.so).  Needs the reference checkout.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as orc  # noqa: E402
from helpers import oracle_spath_run  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    orc.build()
    sc = scenes.icosphere_scene(2)
    blob, counts, bounds = orc.ref_kd_blob(sc.verts4, sc.idx4)
    kt = orc.KdTree(sc.verts4, sc.idx4)
    mask = orc.blob_significant_mask(kt.blob, kt.blob_used)[:blob.shape[0]]
    np.savez_compressed(os.path.join(HERE, "kd_ico2.npz"), ref_blob=blob, mask=mask, counts=counts, bounds=bounds,
                        scene_sha256=hashlib.sha256(sc.verts4.tobytes() + sc.idx4.tobytes()).hexdigest())

    sc = scenes.icosphere_scene(3)
    tree = orc.KdTree(sc.verts4, sc.idx4)
    st = oracle_spath_run(orc, sc, tree, 48, 32, 11, 6)
    np.savez_compressed(os.path.join(HERE, "spath_ico3.npz"), source_results=st.source_results,
                        out_tex=st.out_tex, progress=st.progress, rays=st.rays, image=st.resolve(),
                        scene_sha256=hashlib.sha256(sc.verts4.tobytes() + sc.idx4.tobytes()).hexdigest())
    sc = scenes.sphere_plane_scene()
    rays = orc.generate_rays(96, 64)
    np.savez_compressed(os.path.join(HERE, "ss_sphere_plane.npz"), image=orc.cast_ray_test(sc, rays),
                        first_sample=orc.path_trace(sc, rays, 96, 64, (0, 0, 0), 1))

    cases = obj_cases()
    arrays = {}
    for name, text in cases.items():
        got = orc.ref_obj_parse(text)
        arrays[name + "/text"] = np.frombuffer(text, np.uint8)
        for k, v in got.items():
            arrays[name + "/" + k] = v
    np.savez_compressed(os.path.join(HERE, "obj_cases.npz"), **arrays)
    print("golden fixtures written")


def obj_cases():
    """name -> OBJ file bytes"""
    tri = b"v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0.5\nvn 0 0 1\nvn 0 1 0\nvt 0.25 0.75\nvt 1 0\n"
    return {
        "plain": tri + b"f 1 2 3\nf 2 4 3\n",
        "slashes": tri + b"f 1/1/1 2/2/1 3/1/2\nf 2//2 4//1 3//1\nf 1/2 2/1 4/2\n",
        "negative": tri + b"f -4 -3 -2\nv 2 2 2\nf -1 -2 -3\nf -1/-1/-1 1/1/1 2/2/2\n",
        "quad_fan": tri + b"v 2 0 0\nv 2 1 0\nf 1 2 4 3\nf 1 2 5 6 4\n",
        "crlf": tri.replace(b"\n", b"\r\n") + b"f 1 2 3\r\nf 2 4 3\r\n",
        "lone_cr": b"v 0 0 0\rv 1 0 0\rv 0 1 0\rf 1 2 3\r\n",
        "no_final_newline": tri + b"f 1 2 3\nf 2 4 3",
        "groups": b"g first\n" + tri + b"f 1 2 3\ng second\nf 2 4 3\nf 1 2 4\no third\ng empty\nf 1 3 4\n",
        "faces_before_group": tri + b"f 1 2 3\ng late\nf 2 4 3\ng later\nf 1 2 4\n",
        "group_at_end": tri + b"f 1 2 3\ng trailing\n",
        "numbers": b"v +1.5 -0.25 10\nv 1e2 2.5E1 -3e0\nv 0.1 0.2 0.3\nv 123456.789 0.000123 -0\nv .5 abc 7\n"
                   b"v 1.0000001 16777217 3.4028235\nvn 0.577350 0.577350 0.577350\nf 1 2 3\n",
        "comments_blank": b"# header\n\n  \t\n" + tri + b"   f 1 2 3\n#f 2 4 3\nusemtl thing\nmtllib nothing.mtl\ns off\nf 2 4 3\n",
        "tabs": b"v\t0\t0\t0\nv\t1 0\t0\nv 0 1 0\nf\t1\t2\t3\n",
    }


if __name__ == "__main__":
    main()
