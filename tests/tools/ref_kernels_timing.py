#!/usr/bin/env python3
"""The reference's own kernels timed on the MI355X, beside the library.

    python tests/tools/ref_kernels_timing.py [iterations] [flavour] [c2|c3]

Runs the split path tracer's render loop (xorshift_batch -> kdtree_intersection -> segmented_path_trace per
iteration, src/spath_raytracer.c:355-402) on the code objects compiled from the reference's kernel files
(oracle/_ref/, see oracle/build_ref_kernels.py) at 1024x1024 on the displaced icosphere of 81,920 triangles, and
prints the time of each kernel per iteration and the rate in Mrays/s; c3 is the million-triangle workload at
2048x2048 (tree built by the library, whose serialised form is the reference's wire format).  The reference's
traversal has a private stack of 16 entries without a bounds check (src/kernels/kdtree.cl:78,120); a handful of
rays per million need 17 on c2, and up to 21 on c3: before every traversal launch the rays are checked with the
oracle and the ones that would overflow are replaced by null rays (counted in the output; their pixels are then
wrong, which a timing run can live with).  Launch geometry of the traversal: 2048 one-wave workgroups (the reference launches
num_cores*16 work items in groups of the SIMT width, src/spath_raytracer.c:162-163, with num_cores unset for AMD).
Times are host wall-clock around each dispatch (tens of microseconds of launch cost included).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle                       # noqa: E402
from oracle import refcl            # noqa: E402
from raytracer_amd import scenes    # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    flavour = sys.argv[2] if len(sys.argv) > 2 else "default"
    workload = sys.argv[3] if len(sys.argv) > 3 else "c2"
    groups = int(os.environ.get("REFCL_GROUPS", "2048"))
    if workload == "c3":
        import raytracer_amd as rt
        w = h = 2048
        sc = scenes.icosphere_scene(8)
        host = rt.HostKdTree(sc)                       # the library's builder: same tree, seconds instead of minutes
        tree = oracle.KdTree.from_blob(host.blob, sc.verts4, sc.idx4, host.bounds_min, host.bounds_max)
        tag, name = "ico8", "c3: 1,310,720-triangle displaced icosphere + room, 2048x2048"
    else:
        w = h = 1024
        sc = scenes.icosphere_scene(6)
        tree = oracle.KdTree(sc.verts4, sc.idx4)
        tag, name = "ico6", "c2: 81,920-triangle displaced icosphere + room, 1024x1024"
    n = w * h
    prog = refcl.RefProgram(tag, flavour)
    rs = refcl.RefScene(prog, sc, tree.blob)
    g = refcl.RefSpath(rs, w, h)
    rb = scenes.splitmix32(1, np.arange(n, dtype=np.uint64))
    rv = scenes.splitmix32(2, np.arange(iters + 2, dtype=np.uint64))
    g.prepass(rb, int(rv[0]))
    kd, tr, xs = [], [], []
    t_wall = 0.0
    nulled = 0
    for k in range(iters + 1):
        rays = g.rays
        deep = tree.ray_stack(rays) > 16
        if deep.any():
            rays[deep] = 0
            g.d_rays.write(rays)
            nulled += int(deep.sum())
        t0 = time.perf_counter()
        xs.append(prog.launch("xorshift_batch", [g.d_random], n))
        kd.append(rs.kd_intersect(g.d_results, g.d_rays, n, groups=groups))
        tr.append(prog.launch("segmented_path_trace",
                              [g.d_out_tex, g.d_rays, g.d_origin_rays, g.d_results, g.d_source_results, g.d_progress,
                               g.d_random, rs.materials, rs.meshes, rs.indices, rs.vertices, rs.normals, w,
                               int(rv[k + 1])], n))
        if k > 0:                       # the first iteration traces the wiped (null) rays
            t_wall += time.perf_counter() - t0
    done = len(kd) - 1
    if done < 1:
        return
    kd_ms, tr_ms, xs_ms = float(np.mean(kd[1:])), float(np.mean(tr[1:])), float(np.mean(xs[1:]))
    out = {"workload": name, "flavour": flavour,
           "iterations": done, "traversal_groups": groups, "rays_replaced_by_null_rays": nulled,
           "kdtree_intersection_ms": round(kd_ms, 3), "segmented_path_trace_ms": round(tr_ms, 3),
           "xorshift_batch_ms": round(xs_ms, 3),
           "Mrays_per_s_kernels": round(n / ((kd_ms + tr_ms + xs_ms) * 1e-3) / 1e6, 1),
           "Mrays_per_s_wall": round(n * done / t_wall / 1e6, 1),
           "frame_mean": float(g.out_tex[:, :3].mean())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
