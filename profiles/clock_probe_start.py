"""Diagnostic build only (-DRT_WAVE_TIMES -DRT_WAVE_MILESTONE_SHIFT=7, library named by RT_LIB_PATH): the first
4096 trips of a launch in steps of 128 trips: shader clock and trip rate, with the cost-sorted pixel order and
with index order.  c3 workload."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import raytracer_amd as rt
sc = rt.scenes.icosphere_scene(8)
W = H = 2048
n = W * H
STEP = 128
for reorder in (True, False):
    r = rt.Renderer(sc, W, H).build()
    r.init_spath(seed=1, resolve_interval=50, quiet=True, schedule=0, reorder=reorder)
    r.spath_prepass(num_iterations=1 << 30)
    r.spath_render(50); r.sync()
    for iters in (20, 200):
        r.spath_render(iters); r.sync()
        raw = r.download(r.spath.contents.cl_pixel_cost_buffer, np.uint32, n + 131072)[n:]
        per = raw[:65536].reshape(-1, 8)[:1024]
        mil = raw[65536:].reshape(1024, 32, 2)
        t0 = raw[:65536].reshape(-1, 8)[:, 0]; t0 = t0[t0 != 0].min()
        out = []
        prev_r, prev_c = per[:, 0], per[:, 4]
        for m in range(32):
            b = mil[:, m]
            ok = b[:, 0] != 0
            dreal = (b[ok, 0] - prev_r[ok]).astype(np.int64)
            dclk = (b[ok, 1] - prev_c[ok]).astype(np.int64)
            at = np.median((b[ok, 0] - t0).astype(np.int64)) / 1e5
            out.append((round(float(at), 2), round(float(np.median(dclk / np.maximum(dreal, 1) * 0.1)), 2),
                        round(float(np.median(STEP / (np.maximum(dreal, 1) / 1e5))), 0)))
            prev_r, prev_c = b[:, 0], b[:, 1]
        print(f"reorder={reorder} {iters} iterations: (ms, GHz, trips/ms/wave) per {STEP} trips:")
        print("   ", out)
        r.L.rt_mem_zero(r.dev, r.spath.contents.cl_pixel_cost_buffer.value + 4 * n if hasattr(r.spath.contents.cl_pixel_cost_buffer, 'value') else r.spath.contents.cl_pixel_cost_buffer + 4 * n, 4 * 131072)
        r.sync()
    r.close()
