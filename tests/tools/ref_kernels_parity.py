#!/usr/bin/env python3
"""A configuration at full size on the reference's own kernels and on the library, same RNG streams: c2 = the
81,920-triangle mesh scene at 1024x1024 (64 samples x 4 bounces = 256 iterations), c3 = the 1,310,720-triangle
scene at 2048x2048.

    python tests/tools/ref_kernels_parity.py [iterations] [flavour] [c2|c3]

The reference's kernels (oracle/_ref code objects, launched by oracle/refcl.py) run free, iteration by iteration;
before each traversal launch the oracle checks every ray's stack depth and the few that would overflow the
reference's unchecked 16-entry stack are replaced by null rays (their pixels are listed and left out of the
comparison).  The library renders the same frame through its C API.  Printed: radiance RMSE over the compared
pixels (linear, per channel, averaged image), the share of pixels equal to 1e-5 and to 1e-3, the resolved frames'
8-bit agreement, and both rates.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle                       # noqa: E402
import raytracer_amd as rt          # noqa: E402
from oracle import refcl            # noqa: E402
from raytracer_amd import scenes    # noqa: E402


def compare(iters=256, flavour="default", workload="c2", seed=1, progress=True):
    """Runs both sides and returns the figures (a dict)."""
    if workload == "c3":
        w = h = 2048
        sc = scenes.icosphere_scene(8)
        host = rt.HostKdTree(sc)                       # the library's builder: same tree, seconds instead of minutes
        tree = oracle.KdTree.from_blob(host.blob, sc.verts4, sc.idx4, host.bounds_min, host.bounds_max)
        tag, name = "ico8", "c3: 1,310,720 triangles, 2048x2048"
    else:
        w = h = 1024
        sc = scenes.icosphere_scene(6)
        tree = oracle.KdTree(sc.verts4, sc.idx4)
        tag, name = "ico6", "c2: 81,920 triangles, 1024x1024"
    n = w * h
    rs = refcl.RefScene(refcl.RefProgram(tag, flavour), sc, tree.blob)
    g = refcl.RefSpath(rs, w, h)
    rb = scenes.splitmix32(seed, np.arange(n, dtype=np.uint64))
    rv = scenes.splitmix32(seed + 1, np.arange(iters + 1, dtype=np.uint64))
    g.prepass(rb, int(rv[0]))
    touched = np.zeros(n, bool)
    t_ref = 0.0
    for k in range(iters):
        rays = g.rays
        deep = tree.ray_stack(rays) > 16
        if deep.any():
            rays[deep] = 0
            g.d_rays.write(rays)
            touched |= deep
        t0 = time.perf_counter()
        g.iterate(int(rv[k + 1]))
        t_ref += time.perf_counter() - t0
        if progress and (k + 1) % 32 == 0:
            print(f"reference kernels: iteration {k + 1}", flush=True)
    ref_tex = g.out_tex
    ref_img8 = g.resolve()

    r = rt.Renderer(sc, w, h).build()
    r.init_spath(seed=seed, resolve_interval=50, quiet=True)
    r.spath_prepass(iters)
    r.sync()
    t0 = time.perf_counter()
    r.spath_render(iters)
    r.sync()
    t_lib = time.perf_counter() - t0
    lib_tex = r.spath_state()["out_tex"]
    lib_img8 = r.spath_resolve()
    r.close()

    keep = ~touched
    a = ref_tex[keep, :3] / np.maximum(ref_tex[keep, 3:4], 1)
    b = lib_tex[keep, :3] / np.maximum(lib_tex[keep, 3:4], 1)
    d = np.abs(a - b).max(axis=1)
    ch = lambda x, s: ((x >> s) & 0xff).astype(np.int32)
    lv = np.max([np.abs(ch(ref_img8, s) - ch(lib_img8, s)) for s in (0, 8, 16)], axis=0)[keep]
    out = {"workload": name, "iterations": iters, "flavour": flavour,
           "pixels_left_out": int(touched.sum()),
           "radiance_rmse": float(np.sqrt(np.mean((a - b) ** 2))),
           "image_mean_ref": float(a.mean()), "image_mean_lib": float(b.mean()),
           "pixels_within_1e-5": float((d < 1e-5).mean()), "pixels_within_1e-3": float((d < 1e-3).mean()),
           "samples_equal": float((ref_tex[keep, 3] == lib_tex[keep, 3]).mean()),
           "resolved_pixels_equal": float((lv == 0).mean()), "resolved_within_1_level": float((lv <= 1).mean()),
           "ref_kernels_Mrays_per_s": round(n * iters / t_ref / 1e6, 1),
           "library_Mrays_per_s": round(n * iters / t_lib / 1e6, 1)}
    return out


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    flavour = sys.argv[2] if len(sys.argv) > 2 else "default"
    workload = sys.argv[3] if len(sys.argv) > 3 else "c2"
    print(json.dumps(compare(iters, flavour, workload)))


if __name__ == "__main__":
    main()
