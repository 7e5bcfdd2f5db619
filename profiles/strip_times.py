"""Time each rank's share of the c3 frame on one GPU, for contiguous strips or (argument: bands) interleaved bands.
STEPS (environment, default 100) iterations are timed after 50 of warm-up; they run as one launch."""
import os
STEPS = int(os.environ.get("STEPS", "100"))
import sys, time
sys.path.insert(0, '.')
import raytracer_amd as rt
from raytracer_amd.strips import band_partition
BANDS = len(sys.argv) > 1 and sys.argv[1] == 'bands'
sc = rt.scenes.icosphere_scene(8)
W = H = 2048
for N in ((8,) if len(sys.argv) > 2 else (2, 4, 8)):
    rows = H // N
    ts, picked = [], []
    for k in range(N if len(sys.argv) <= 2 else 2):
        if BANDS:
            first, band, stride, rows = band_partition(H, N, k)
            r = rt.Renderer(sc, W, rows, tile_y0=first, full_height=H, bands=(band, stride)).build()
        else:
            r = rt.Renderer(sc, W, rows, tile_y0=k * rows, full_height=H).build()
        r.init_spath(seed=1, resolve_interval=50, quiet=True)
        r.spath_prepass(num_iterations=1 << 30)
        r.spath_render(50); r.sync()
        t0 = time.perf_counter(); r.spath_render(STEPS); r.sync(); ts.append(time.perf_counter() - t0)
        picked.append(int(r.spath.contents.schedule))
        r.close()
    full = STEPS * W * H / 1e6
    print(N, [round(t * 1e3, 1) for t in ts], "max-based Mrays/s", round(full / max(ts)), "ideal(sum)", round(full / sum(ts) * N), "schedules", picked, flush=True)
