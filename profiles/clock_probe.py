"""Diagnostic build only (-DRT_WAVE_TIMES, library named by RT_LIB_PATH): the shader clock (s_memtime ticks per
100 MHz s_memrealtime tick) and the trip rate of k_spath_run's waves as a launch goes on, and how both depend on
what the GPU did just before the launch.  c3 workload, default schedule fixed."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import raytracer_amd as rt
sc = rt.scenes.icosphere_scene(8)
W = H = 2048
n = W * H
r = rt.Renderer(sc, W, H).build()
r.init_spath(seed=1, resolve_interval=50, quiet=True, schedule=0)
r.spath_prepass(num_iterations=1 << 30)


def stamps():
    raw = r.download(r.spath.contents.cl_pixel_cost_buffer, np.uint32, n + 131072)[n:]
    per = raw[:65536].reshape(-1, 8)
    per = per[per[:, 1] != 0]
    mil = raw[65536:].reshape(1024, 32, 2)
    return per, mil


def report(tag):
    per, mil = stamps()
    real = lambda a, b: (per[:, b] - per[:, a]).astype(np.int64)
    t0 = per[:, 0].min()
    end = ((per[:, 1] - t0).astype(np.int64) / 1e5)
    dry = ((per[:, 2] - t0).astype(np.int64) / 1e5)
    bulk = real(4, 6) / np.maximum(real(0, 2), 1) * 0.1
    drain_ok = real(2, 1) > 1000
    drain = (real(6, 5) / np.maximum(real(2, 1), 1) * 0.1)[drain_ok]
    print(f"{tag}: launch {end.max():.2f} ms, dry at {np.median(dry):.2f} ms, clock bulk {np.median(bulk):.3f} GHz, "
          f"drain {np.median(drain):.3f} GHz, trips/wave {np.median(per[:, 3]):.0f}")
    return per, mil


def trajectory(per, mil, tag):
    """clock and trips per ms between consecutive 1024-trip milestones, median over the first 1024 waves"""
    t0 = per[:, 0].min()
    out = []
    for m in range(1, 32):
        a, b = mil[:, m - 1], mil[:, m]
        ok = (b[:, 0] != 0) & (a[:, 0] != 0)
        if ok.sum() < 512:
            break
        dreal = (b[ok, 0] - a[ok, 0]).astype(np.int64)
        dclk = (b[ok, 1] - a[ok, 1]).astype(np.int64)
        at = np.median((b[ok, 0] - t0).astype(np.int64)) / 1e5
        out.append((round(at, 1), round(float(np.median(dclk / dreal * 0.1)), 3), round(float(np.median(1024 / (dreal / 1e5))), 1)))
    print(f"{tag}: (ms into the launch, GHz, trips per ms per wave) per 1024 trips:")
    print("   ", out)


r.spath_render(50); r.sync()
# E1: one long launch after a sync
r.spath_render(200); r.sync()
per, mil = report("E1 200 iterations after a sync")
trajectory(per, mil, "E1")
# E2: a 20-iteration launch after a sync (the benchmark's short timed region)
time.sleep(0.2)
r.spath_render(20); r.sync()
per, mil = report("E2 20 iterations after 0.2 s of idling")
trajectory(per, mil, "E2")
r.spath_render(20); r.sync()
report("E2b 20 iterations right after a synchronised 20-iteration launch")
# E3: a 20-iteration launch queued directly behind a 200-iteration launch, no host synchronisation between them
r.spath_render(200, flush=True)
r.spath_render(20); r.sync()
per, mil = report("E3 20 iterations queued behind a 200-iteration launch (no sync between)")
trajectory(per, mil, "E3")
# E4: five 20-iteration launches queued back to back; the last one's stamps survive
for k in range(5):
    r.spath_render(20, flush=True)
r.sync()
report("E4 the fifth of five 20-iteration launches queued back to back")
# E5: 200 iterations as ten 20-iteration launches vs one launch: host time
for batch in (20, 200):
    r.spath.contents.max_batch = batch
    r.sync(); t = time.perf_counter()
    r.spath_render(200); r.sync()
    print(f"E5 200 iterations in launches of {batch}: {1e3 * (time.perf_counter() - t):.2f} ms")
