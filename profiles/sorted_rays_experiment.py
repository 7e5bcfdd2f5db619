"""What ray coherence is worth to the traversal kernel: the bounce rays a render holds after a few iterations are
traversed by rt_k_kd_intersect (one ray per lane, persistent waves, consecutive rays to consecutive lanes) in
pixel order and in sorted orders -- sorted on the host here, so only the traversal's gain is measured, not a
device sort's cost -- and the hit records are compared (a ray's walk depends on nothing else: they must be equal).

    python profiles/sorted_rays_experiment.py <workload> [iterations before the snapshot]

Keys: Morton code of the ray's origin in the scene box (10 bits per axis) with the direction's octant (3 sign bits)
on top ("octant|morton"), the Morton code alone, and the octant below a coarser Morton code (5 bits per axis on
top, then octant, then the fine bits).  Controls: the same rays in a random order, and the frame's primary
rays (fully coherent) in pixel order and shuffled."""
import ctypes as C
import json
import sys
import time
sys.path.insert(0, '.')
import numpy as np
import raytracer_amd as rt
from bench import WORKLOADS, make_scene

wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 6
spec, W, H, desc = WORKLOADS[wl]
sc = make_scene(spec)
r = rt.Renderer(sc, W, H).build()
r.init_spath(seed=1, resolve_interval=0, quiet=True, schedule=0)
r.spath_prepass(num_iterations=1 << 30)
r.spath_render(warm); r.sync()
n = W * H
b = r.spath_buffers()
ro = r.download(b.ray_o, np.float32, 4 * n).reshape(n, 4)
rd = r.download(b.ray_d, np.float32, 4 * n).reshape(n, 4)
lo = np.array(r.kdt.bounds.min[:3], np.float32)
hi = np.array(r.kdt.bounds.max[:3], np.float32)


def part1by2(x):
    x = x.astype(np.uint64) & 0x3ff
    x = (x | (x << 16)) & 0x30000ff
    x = (x | (x << 8)) & 0x300f00f
    x = (x | (x << 4)) & 0x30c30c3
    x = (x | (x << 2)) & 0x9249249
    return x


cell = np.clip(((ro[:, :3] - lo) / (hi - lo) * 1024).astype(np.int64), 0, 1023)
morton = part1by2(cell[:, 0]) | (part1by2(cell[:, 1]) << 1) | (part1by2(cell[:, 2]) << 2)
octant = ((rd[:, 0] < 0).astype(np.uint64) | ((rd[:, 1] < 0).astype(np.uint64) << 1) | ((rd[:, 2] < 0).astype(np.uint64) << 2))
keys = {
    "pixel order": None,
    "octant|morton30": (octant << 30) | morton,
    "morton30": morton,
    "morton15|octant|morton15": ((morton >> 15) << 18) | (octant << 15) | (morton & 0x7fff),
    # control: the same rays in a random order -- what the traversal loses when there is no coherence at all
    "random shuffle": np.random.default_rng(7).permutation(n).astype(np.uint64),
}
L, dev, view = r.L, r.dev, C.byref(r.kdt.view)
d_o = L.rt_mem_alloc(dev, 16 * n); d_d = L.rt_mem_alloc(dev, 16 * n); d_h = L.rt_mem_alloc(dev, 16 * n)
base = None
out = {"workload": f"{wl}: {desc}", "rays": n, "bounce_rays_after_iterations": warm, "orders": {}}
for name, key in keys.items():
    perm = np.arange(n) if key is None else np.argsort(key, kind="stable")
    L.rt_mem_upload(dev, d_o, np.ascontiguousarray(ro[perm]).ctypes.data_as(C.c_void_p), 16 * n)
    L.rt_mem_upload(dev, d_d, np.ascontiguousarray(rd[perm]).ctypes.data_as(C.c_void_p), 16 * n)
    ms = []
    for rep in range(4):
        r.timing(True); r.timing_reset()
        L.rt_k_kd_intersect(dev, view, d_o, d_d, d_h, n)
        r.sync()
        ms.append(r.timing_get("kd_intersect")[1])
        r.timing(False)
    hits = r.download(d_h, np.uint32, 4 * n).reshape(n, 4)
    if base is None:
        base = hits
    same = bool((hits == base[perm]).all())
    out["orders"][name] = {"ms": [round(x, 3) for x in ms], "best_ms": round(min(ms[1:]), 3),
                           "Mrays_per_s": round(n / min(ms[1:]) / 1e3, 1), "hit_records_equal_to_pixel_order": same}
    print(name, out["orders"][name], flush=True)
# control: primary rays (the saved origin rays), coherent in pixel order, and the same rays shuffled
oo = r.download(b.origin_o, np.float32, 4 * n).reshape(n, 4)
od = r.download(b.origin_d, np.float32, 4 * n).reshape(n, 4)
for name, perm in (("primary rays, pixel order", np.arange(n)), ("primary rays, random shuffle", np.random.default_rng(7).permutation(n))):
    L.rt_mem_upload(dev, d_o, np.ascontiguousarray(oo[perm]).ctypes.data_as(C.c_void_p), 16 * n)
    L.rt_mem_upload(dev, d_d, np.ascontiguousarray(od[perm]).ctypes.data_as(C.c_void_p), 16 * n)
    ms = []
    for rep in range(4):
        r.timing(True); r.timing_reset()
        L.rt_k_kd_intersect(dev, view, d_o, d_d, d_h, n)
        r.sync()
        ms.append(r.timing_get("kd_intersect")[1])
        r.timing(False)
    out["orders"][name] = {"ms": [round(x, 3) for x in ms], "best_ms": round(min(ms[1:]), 3), "Mrays_per_s": round(n / min(ms[1:]) / 1e3, 1)}
    print(name, out["orders"][name], flush=True)
p = {k: v for k, v in out["orders"].items() if not k.startswith("primary") and k != "random shuffle"}
out["traversal_speedup_of_best_sorted_order"] = round(p["pixel order"]["best_ms"] / min(v["best_ms"] for k, v in p.items() if k != "pixel order"), 3)
print(json.dumps(out))
