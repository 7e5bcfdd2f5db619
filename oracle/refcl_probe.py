#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: staged bring-up of the reference-kernel launcher on a GPU box.

    python oracle/refcl_probe.py [stage]   # stages run in order up to `stage` (default: all)

Each stage checks its result before the next one touches the GPU; the first failure ends the run.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle                       # noqa: E402
from oracle import refcl            # noqa: E402
from raytracer_amd import scenes    # noqa: E402


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7fffffff), a)
    b = np.where(b < 0, -(b & 0x7fffffff), b)
    return np.abs(a - b)


def main():
    last = int(sys.argv[1]) if len(sys.argv) > 1 else 99
    flavour = os.environ.get("REFCL_FLAVOUR", "strict")
    prog = refcl.RefProgram("ico4", flavour)
    print("loaded", prog.meta["tag"], flavour, flush=True)

    # 1: a kernel with one plain buffer argument
    n = 4096
    data = (np.arange(n, dtype=np.uint64) * 2654435761 % (1 << 32)).astype(np.uint32)
    d = refcl.DevBuf.of(data.astype(np.uint32))
    prog.launch("xorshift_batch", [d], n)
    want = data.astype(np.uint32).copy()
    oracle.xorshift(want)
    got = d.read(np.uint32, n)
    assert (got == want).all(), "xorshift_batch differs"
    print("stage 1 ok: xorshift_batch == oracle", flush=True)
    if last < 2:
        return

    # 2: by-value arguments of several sizes
    sc = scenes.icosphere_scene(4)
    tree = oracle.KdTree(sc.verts4, sc.idx4)
    rs = refcl.RefScene(prog, sc, tree.blob)
    w = h = 64
    rays = refcl.DevBuf.zeros(32 * w * h)
    rs.generate_rays(rays, w, h)
    got = rays.read(np.float32, 8 * w * h).reshape(-1, 8)
    want = oracle.generate_rays(w, h)
    u = ulp_diff(got, want)
    print("stage 2: generate_rays max ulp diff vs oracle", int(u.max()), "words differing", int((u != 0).sum()), flush=True)
    assert u.max() <= 4
    if last < 3:
        return

    # 3: image-buffer arguments + private stack: traversal of the primary rays
    _, st = tree.intersect(want, stats=True)
    assert st["max_stack"] <= 16, "the reference's 16-entry stack would overflow"
    hits = refcl.DevBuf.zeros(16 * w * h)
    rays.write(want)
    ms = rs.kd_intersect(hits, rays, w * h)
    got = hits.read(refcl.HIT_DT, w * h)
    oracle.set_builtins("plain")
    exp = tree.intersect(want)
    oracle.set_builtins("rocm")
    same = got["tri"] == exp["tri"]
    print(f"stage 3: kdtree_intersection {ms:.2f} ms, ids equal {int(same.sum())}/{w * h}, hits {int((exp['t'] != 0).sum())}",
          flush=True)
    for f in ("t", "u", "v"):
        print("   ", f, "max ulp diff", int(ulp_diff(got[f][same], exp[f][same]).max()),
              "bit-equal", int((got[f].view(np.uint32) == exp[f].view(np.uint32)).sum()), flush=True)
    assert same.mean() > 0.999
    exp2 = tree.intersect(want)
    print("    against the oracle (default: ROCm's dot/cross/min/max): words differing",
          int((got.view(np.uint32) != exp2.view(np.uint32)).sum()), "of", 4 * w * h, flush=True)
    if last < 4:
        return

    # 4: the split path tracer's sequence
    w, h, iters, seed = 64, 64, 6, 3
    rb = scenes.splitmix32(seed, np.arange(w * h, dtype=np.uint64))
    rv = scenes.splitmix32(seed + 1, np.arange(iters + 1, dtype=np.uint64))
    o = oracle.SpathState(sc, tree, w, h)
    g = refcl.RefSpath(rs, w, h)
    o.prepass(rb, int(rv[0]))
    g.prepass(rb, int(rv[0]))
    print("stage 4: first-bounce rays max ulp", int(ulp_diff(g.first_bounce_rays, o.first_bounce_rays).max()), flush=True)
    for k in range(iters):
        o.iterate(int(rv[k + 1]))
        g.iterate(int(rv[k + 1]))
        go, oo = g.out_tex, o.out_tex
        ids = (g.results["tri"] == o.results["tri"]).mean()
        print(f"   iteration {k + 1}: ids equal {ids:.4f}, out_tex max abs diff {np.abs(go - oo).max():.3e}", flush=True)
    img_g, img_o = g.resolve(), o.resolve()
    print("   resolved pixels equal", int((img_g == img_o).sum()), "of", w * h, flush=True)


if __name__ == "__main__":
    main()
