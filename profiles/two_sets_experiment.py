"""Experiment: does a second, independent set of pixels on another stream fill the GPU while the first set's launch
drains?  The c3 frame is rendered (a) by one context, 200 iterations in launches of SUB iterations, and (b) by two
contexts that hold the even and the odd 8-row bands of the frame (the two-rank partition) on one GPU, each with its own
streams, their launches of SUB iterations queued alternately.  Wall time for the same 200 iterations of the frame."""
import sys, time
sys.path.insert(0, '.')
import raytracer_amd as rt
from raytracer_amd.strips import band_partition
sc = rt.scenes.icosphere_scene(8)
W = H = 2048
ITERS = 200

def one(sub):
    r = rt.Renderer(sc, W, H).build()
    r.init_spath(seed=1, resolve_interval=0, quiet=True, schedule=0, max_batch=sub)
    r.spath_prepass(num_iterations=1 << 30)
    r.spath_render(50); r.sync()
    t0 = time.perf_counter(); r.spath_render(ITERS); r.sync(); dt = time.perf_counter() - t0
    r.close()
    return dt

def two(sub):
    rs = []
    for k in range(2):
        first, band, stride, rows = band_partition(H, 2, k)
        r = rt.Renderer(sc, W, rows, tile_y0=first, full_height=H, bands=(band, stride)).build()
        r.L.rt_device_select_stream(r.dev, k)        # streams are dealt to the hardware queues in turn: take different ones
        r.init_spath(seed=1, resolve_interval=0, quiet=True, schedule=0, max_batch=sub)
        r.spath_prepass(num_iterations=1 << 30)
        rs.append(r)
    for r in rs:
        r.spath_render(50)
    for r in rs:
        r.sync()
    t0 = time.perf_counter()
    for j in range(0, ITERS, sub):
        for r in rs:
            r.spath_render(min(sub, ITERS - j))       # queues one launch, does not wait
    for r in rs:
        r.sync()
    dt = time.perf_counter() - t0
    for r in rs:
        r.close()
    return dt

for sub in (200, 50, 25, 10):
    a = one(sub); b = two(sub)
    rays = ITERS * W * H / 1e6
    print(f"launches of {sub:3d}: one context {a*1e3:7.1f} ms ({rays/a:6.0f} Mrays/s)   two interleaved sets {b*1e3:7.1f} ms ({rays/b:6.0f} Mrays/s)", flush=True)
