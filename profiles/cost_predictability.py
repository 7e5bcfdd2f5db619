"""How well does a pixel's walk cost in one launch predict its cost in the next (c3 frame)?  The heaviest-first order
of a launch is built from the costs the launch before it recorded; a launch ends when its slowest pixel does.
usage: cost_predictability.py [iterations per launch, default 20]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import raytracer_amd as rt
IT = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sc = rt.scenes.icosphere_scene(8)
W = H = 2048
r = rt.Renderer(sc, W, H).build()
r.init_spath(seed=1, resolve_interval=50, quiet=True)
r.spath_prepass(num_iterations=1 << 30)
r.spath_render(5); r.sync()
cost = []
for k in range(3):
    r.spath_render(IT); r.sync()
    cost.append(r.download(r.spath.contents.cl_pixel_cost_buffer, np.uint32, W * H).astype(np.float64))
a, b = cost[1], cost[2]
print(f"{IT} iterations per launch: mean cost {a.mean():.0f} trips, p99 {np.percentile(a, 99):.0f}, max {a.max():.0f} ({a.max() / a.mean():.1f} x mean)")
print("correlation of a pixel's cost in consecutive launches: %.3f" % np.corrcoef(a, b)[0, 1])
tiles = lambda c: c.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
ta, tb = tiles(a), tiles(b)
print("correlation of a tile's mean cost in consecutive launches: %.3f" % np.corrcoef(ta.mean(1), tb.mean(1))[0, 1])
print("correlation of a tile's max cost in consecutive launches:  %.3f" % np.corrcoef(ta.max(1), tb.max(1))[0, 1])
top = np.argsort(a)[-len(a) // 100:]
print("the heaviest 1%% of pixels of one launch: mean cost %.0f, in the next launch %.0f (frame mean %.0f); share among the next launch's heaviest 1%%: %.2f"
      % (a[top].mean(), b[top].mean(), b.mean(), np.isin(top, np.argsort(b)[-len(b) // 100:]).mean()))
ratio = tb.max(1) / np.maximum(tb.mean(1), 1)
print("max / mean cost inside a tile: median %.2f, p90 %.2f, p99 %.2f" % tuple(np.percentile(ratio, [50, 90, 99])))
# the pixels that end latest if every tile started in the order of the previous launch's tile means and ran alone:
order = np.argsort(-ta.mean(1), kind="stable")
lanes = 7168 * 64
start = np.repeat(np.cumsum(tb.mean(1)[order]) * 64 / lanes, 1)            # when a tile's turn comes (in trips of work per lane)
finish_tile = start + tb.max(1)[order]
print("ideal length %.0f trips per lane; last tile's pixels would end at %.0f, the latest single pixel at %.0f (+%.1f %%)"
      % (b.sum() / lanes, (start + tb.mean(1)[order]).max(), finish_tile.max(), 100 * (finish_tile.max() / (b.sum() / lanes) - 1)))
late = np.argsort(-finish_tile)[:5]
print("tiles that end last: position in the order (of %d) / mean / max:" % len(order), [(int(i), round(float(tb.mean(1)[order][i])), round(float(tb.max(1)[order][i]))) for i in late])
