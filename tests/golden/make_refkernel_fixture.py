#!/usr/bin/env python3
"""Writes refkernel_ico4.npz and refkernel_brute.npz: inputs and outputs of the REFERENCE's own kernels, run on an
MI355X.

Needs a GPU box and oracle/_ref/refcl_ico4_strict.hsaco (the reference's OpenCL C kernel files compiled, unmodified,
for gfx950 by oracle/build_ref_kernels.py where the reference checkout is mounted; flavour "strict": no contraction,
correctly rounded division and square root).  The kernels are launched by oracle/refcl.py with the reference's
argument lists.  What is frozen (scene: raytracer_amd.scenes.icosphere_scene(4), 64x64 pixels, RNG streams of seed 5):

  rays / hits              kdtree_intersection: 4096 primary rays, 4096 wiped (null) rays, 3 x 4096 bounce rays and
                           2048 hand-made rays (exact zeros in the direction) with the hit records it returned
  init_*                   segmented_path_trace_init: the primary hits going in, the buffers coming out
  step_pre_* / step_post_* segmented_path_trace (after xorshift_batch), iterations 2 and 5: the buffers going in, with
                           the hit records of that iteration's traversal, and the buffers coming out
  xorshift_in / _out       xorshift_batch

The file goes to gpurun_out/ (the only directory a GPU run hands back); move it to tests/golden/ by hand.
    python tests/golden/make_refkernel_fixture.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle                       # noqa: E402
from oracle import refcl            # noqa: E402
from raytracer_amd import scenes    # noqa: E402


def main():
    w = h = 64
    n = w * h
    seed, iters = 5, 5
    sc = scenes.icosphere_scene(4)
    tree = oracle.KdTree(sc.verts4, sc.idx4)
    prog = refcl.RefProgram("ico4", "strict")
    rs = refcl.RefScene(prog, sc, tree.blob)
    g = refcl.RefSpath(rs, w, h)
    rb = scenes.splitmix32(seed, np.arange(n, dtype=np.uint64))
    rv = scenes.splitmix32(seed + 1, np.arange(iters + 1, dtype=np.uint64))
    out = {"width": w, "height": h, "seed": seed, "random_values": rv,
           "scene_sha256": hashlib.sha256(sc.verts4.tobytes() + sc.idx4.tobytes()).hexdigest(),
           "build_flags": " ".join(prog.meta["flags"])}

    g.prepass(rb, int(rv[0]), zero_rays=False)
    ray_sets, hit_sets = [g.primary_rays], [g.source_results]
    out["init_results"] = g.source_results
    out["init_out_tex"], out["init_rays"], out["init_origin_rays"] = g.out_tex, g.rays, g.origin_rays
    out["init_progress"] = g.progress
    g.d_rays.zero()                                     # src/spath_raytracer.c:432-433
    for k in range(iters):
        pre = {"rays": g.rays, "progress": g.progress, "out_tex": g.out_tex, "random": g.random}
        assert tree.ray_stack(pre["rays"]).max() <= 16
        g.iterate(int(rv[k + 1]))
        if k < 4:
            ray_sets.append(pre["rays"]); hit_sets.append(g.results)
        if k == 0:
            out["xorshift_in"], out["xorshift_out"] = pre["random"], g.random
        elif k + 1 in (2, 5):
            for name, a in pre.items():
                out[f"step_pre_{name}_{k + 1}"] = a
            out[f"step_pre_results_{k + 1}"] = g.results
            out[f"step_post_rays_{k + 1}"], out[f"step_post_progress_{k + 1}"] = g.rays, g.progress
            out[f"step_post_out_tex_{k + 1}"], out[f"step_post_random_{k + 1}"] = g.out_tex, g.random
    out["origin_rays"], out["source_results"] = g.origin_rays, g.source_results

    rng = np.random.RandomState(seed)
    m = 2048
    extra = np.zeros((m, 8), np.float32)
    extra[:, 0:3] = rng.uniform(-35, 35, size=(m, 3))
    d = rng.normal(size=(m, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for k in range(m // 2):
        d[k, rng.randint(3)] = (0.0, -0.0, 1.0, -1.0)[rng.randint(4)]
    extra[:, 4:7] = d
    extra = np.ascontiguousarray(extra[tree.ray_stack(extra) <= 16][: m // 64 * 64 - 64])
    d_hits = refcl.DevBuf.zeros(16 * extra.shape[0])
    rs.kd_intersect(d_hits, refcl.DevBuf.of(extra), extra.shape[0])
    ray_sets.append(extra); hit_sets.append(d_hits.read(refcl.HIT_DT, extra.shape[0]))
    out["rays"] = np.concatenate(ray_sets)
    out["hits"] = np.concatenate(hit_sets)

    brute_force_fixture()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "refkernel_ico4.npz")
    np.savez_compressed(path, **out)
    print("written", path, os.path.getsize(path), "bytes;", out["rays"].shape[0], "rays,",
          int((out["hits"]["t"] != 0).sum()), "hits")


def brute_force_fixture():
    """refkernel_brute.npz: cast_ray_test on the sphere + plane scene (128x128) and on the mixed scene (96x64), and
    the first path_trace launch on the mixed scene, from the reference's kernels (flavour strict); the rays going in
    are the oracle's generate_rays output."""
    out = {}
    for tag, sc, w, h in (("spheres", scenes.sphere_plane_scene(), 128, 128), ("mixed2", scenes.mixed_scene(2), 96, 64)):
        tree = oracle.KdTree(sc.verts4, sc.idx4) if sc.idx4.shape[0] else None
        rs = refcl.RefScene(refcl.RefProgram(tag, "strict"), sc, tree.blob if tree is not None else None)
        rays = oracle.generate_rays(w, h)
        d_rays, img = refcl.DevBuf.of(rays), refcl.DevBuf.zeros(4 * w * h)
        rs.cast_ray_test(img, d_rays, w, h)
        out[f"{tag}_size"] = np.array([w, h])
        out[f"{tag}_cast_ray_test"] = img.read(np.uint32, w * h)
        if tag == "mixed2":
            fresh = refcl.DevBuf.zeros(16 * w * h)
            rs.path_trace(fresh, d_rays, w, h, (0.0, 0.0, 0.0), 1)
            out["mixed2_path_trace_first"] = fresh.read(np.float32, 4 * w * h).reshape(-1, 4)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "refkernel_brute.npz"), **out)
    print("written gpurun_out/refkernel_brute.npz")


if __name__ == "__main__":
    main()
