#!/usr/bin/env python3
"""Writes the golden fixtures under tests/golden/.

kd_ico2.npz   serialised kd-tree of scenes.icosphere_scene(2) from the
              REFERENCE's own builder (oracle/_ref, built from /root/reference),
              plus the mask of bytes that carry data (the reference leaves
              struct padding uninitialised).  Needs the reference checkout.
spath_ico3.npz  oracle outputs for a 48x32 render of icosphere_scene(3):
              primary hits, and out_tex / progress after 6 iterations with the
              counter-based RNG streams (seed 11).  These pin the oracle against
              its own earlier output (the reference has no vectors for the
              kernels), so a change in them is a deliberate, reviewed event.
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
    print("golden fixtures written")


if __name__ == "__main__":
    main()
