"""Distribution of the per-pixel walk cost (trips per 50-iteration batch) the render kernel records, c3 frame."""
import sys
sys.path.insert(0, '.')
import numpy as np
import raytracer_amd as rt
sc = rt.scenes.icosphere_scene(8)
W = H = 2048
r = rt.Renderer(sc, W, H).build()
r.init_spath(seed=1, resolve_interval=50, quiet=True)
r.spath_prepass(num_iterations=1 << 30)
r.spath_render(100); r.sync()
c = r.download(r.spath.contents.cl_pixel_cost_buffer, np.uint32, W * H).astype(np.float64)
print("mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  p99.9 %.0f  max %.0f" % (c.mean(), *np.percentile(c, [50, 90, 99, 99.9]), c.max()))
print("share of total work in the top 1%% of pixels: %.3f" % (np.sort(c)[-len(c) // 100:].sum() / c.sum()))
r.timing(True); r.timing_reset(); r.spath_render(50); r.sync()
n, ms = r.timing_get("spath_run")
print("launch of 50 iterations: %.2f ms -> %.2f us per trip of the heaviest pixel if it alone set the time" % (ms / n, ms / n * 1e3 / c.max()))
