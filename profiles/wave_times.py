"""Diagnostic build only (rt_kernels.hip compiled with -DRT_WAVE_TIMES, library named by RT_LIB_PATH): when do
the waves of one launch of ITERS iterations (argv[1], default 50) start, see the work queue run dry, and exit, and
at which shader clock did they run (s_memtime ticks per 100 MHz s_memrealtime tick)?  WARM (argv[2], default 100)
iterations run before the measured launch.  Times in ms from the first wave's start (100 MHz wall clock)."""
import sys
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 50
WARM = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sys.path.insert(0, '.')
import numpy as np
import raytracer_amd as rt
sc = rt.scenes.icosphere_scene(8)
W = H = 2048
r = rt.Renderer(sc, W, H).build()
r.init_spath(seed=1, resolve_interval=50, quiet=True, schedule=0)
r.spath_prepass(num_iterations=1 << 30)
r.spath_render(WARM); r.sync()
r.spath_render(ITERS); r.sync()
n = W * H
raw = r.download(r.spath.contents.cl_pixel_cost_buffer, np.uint32, n + 65536)[n:].reshape(-1, 8)
raw = raw[(raw[:, 1] != 0)]
t0 = raw[:, 0].min()
start, end, dry, trips = [(raw[:, k] - t0).astype(np.int64) / 1e5 for k in (0, 1, 2)] + [raw[:, 3]]
pct = [0, 1, 10, 50, 90, 99, 100]
print("iterations", ITERS, "after", WARM, "warm-up iterations; waves", len(raw))
print("start  ms", np.percentile(start, pct).round(2))
print("dry    ms", np.percentile(dry, pct).round(2))
print("exit   ms", np.percentile(end, pct).round(2))
print("exit - dry ms", np.percentile(end - dry, pct).round(2))
print("trips", np.percentile(trips, pct).round(0))
alive = [(end > t).sum() for t in np.linspace(0, end.max(), 21)]
print("waves alive at 5% steps of the launch:", alive)
# shader clock: counter ticks per wall tick (100 MHz) over each wave's life, its bulk (start -> dry) and its drain
real = lambda a, b: (raw[:, b] - raw[:, a]).astype(np.int64)
clk = lambda a, b: (raw[:, b] - raw[:, a]).astype(np.int64)
life = clk(4, 5) / np.maximum(real(0, 1), 1) * 0.1
bulk = clk(4, 6) / np.maximum(real(0, 2), 1) * 0.1
drain = clk(6, 5) / np.maximum(real(2, 1), 1) * 0.1
print("shader clock GHz over a wave's life", np.percentile(life, pct).round(3))
print("shader clock GHz start -> dry      ", np.percentile(bulk, pct).round(3))
print("shader clock GHz dry -> exit       ", np.percentile(drain[real(2, 1) > 1000], pct).round(3))
print("trips per ms of the bulk (per wave)", np.percentile(trips / np.maximum(end, 1e-3), pct).round(1))
