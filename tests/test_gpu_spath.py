"""GPU parity: the HIP path through the C API against the CPU oracle, bit for bit."""
import numpy as np
import pytest

from helpers import assert_bits_equal, hits_equal, oracle_spath_run

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small(rt, orc):
    sc = rt.scenes.icosphere_scene(4)
    tree = orc.KdTree(sc.verts4, sc.idx4)
    return sc, tree


def test_generate_rays_bit_exact(rt, orc, small):
    sc, _ = small
    for (w, h) in ((64, 64), (96, 40), (33, 17)):
        r = rt.Renderer(sc, w, h).build()
        r.L._raytracer_gen_ray_buffer(r.rctx)
        n = w * h
        ro = r.download(r.rctx.contents.cl_ray_buffer, np.float32, 4 * n).reshape(n, 4)
        rd = r.download(r.rctx.contents.cl_ray_dir_buffer, np.float32, 4 * n).reshape(n, 4)
        want = orc.generate_rays(w, h)
        assert_bits_equal(ro, want[:, :4], "ray origins")
        assert_bits_equal(rd, want[:, 4:], "ray directions")
        r.close()


@pytest.mark.parametrize("fused", [False, True])  # fused = RNG step folded into the shading launch
def test_spath_state_bit_exact(rt, orc, small, fused):
    sc, tree = small
    w, h, iters, seed = 96, 64, 9, 3
    r = rt.Renderer(sc, w, h).build()
    r.init_spath(seed=seed, separate_kernels=not fused, resolve_interval=0)
    r.spath_prepass(num_iterations=iters)

    # primary hits and the state the init kernel leaves
    st0 = oracle_spath_run(orc, sc, tree, w, h, seed, 0)
    g0 = r.spath_state()
    hits_equal(g0["source_results"], st0.source_results, "primary hits")
    assert (g0["source_results"]["t"] != 0).sum() > 100
    assert_bits_equal(g0["origin_rays"], st0.origin_rays, "origin rays")
    assert_bits_equal(g0["out_tex"], st0.out_tex, "out_tex after init")
    assert_bits_equal(g0["progress"], st0.progress, "progress after init")
    assert_bits_equal(g0["random"], st0.random, "random buffer")
    assert_bits_equal(g0["rays"], st0.rays, "rays after prepass (zeroed)")

    r.spath_render(iters)
    st = oracle_spath_run(orc, sc, tree, w, h, seed, iters)
    g = r.spath_state()
    hits_equal(g["results"], st.results, "bounce hits")
    assert_bits_equal(g["rays"], st.rays, "rays")
    assert_bits_equal(g["progress"], st.progress, "progress")
    assert_bits_equal(g["random"], st.random, "random")
    assert_bits_equal(g["out_tex"], st.out_tex, "out_tex")
    img = r.spath_resolve()
    assert_bits_equal(img, st.resolve(), "resolved ARGB image")
    assert st.out_tex[:, 3].max() >= 2            # at least two finished samples somewhere
    r.close()


def test_first_bounce_rays_bit_exact(rt, orc, small):
    """The rays the init kernel writes (before the reference's prepass wipes them)."""
    sc, tree = small
    w, h, seed = 64, 64, 5
    r = rt.Renderer(sc, w, h).build()
    r.init_spath(seed=seed)
    r.L.raytracer_prepass(r.rctx)
    r.L._raytracer_gen_ray_buffer(r.rctx)
    b = r.spath_buffers()
    n = w * h
    kd = r.kdt.view
    import ctypes as C
    r.L.rt_k_kd_intersect(r.dev, C.byref(kd), b.ray_o, b.ray_d, b.hits, n)
    from raytracer_amd.scenes import splitmix32
    rv = int(splitmix32(seed + 1, np.arange(1, dtype=np.uint64))[0])
    rows = rt.rt_rows(0, 0, 0, h)
    r.L.rt_k_spath_init(r.dev, C.byref(kd), C.byref(b), w, h, C.byref(rows), rv)
    r.sync()
    g = r.spath_state()
    st = orc.SpathState(sc, tree, w, h)
    st.prepass(np.zeros(n, np.uint32), rv, zero_rays=False)
    assert_bits_equal(g["rays"], st.first_bounce_rays, "first bounce rays")
    r.close()


def test_tile_equals_full_frame(rt, orc, small):
    """A row strip rendered on its own gives the same pixels as the full frame."""
    sc, tree = small
    w, h, iters, seed = 64, 48, 6, 7
    full = rt.Renderer(sc, w, h).build()
    full.init_spath(seed=seed, resolve_interval=0)
    full.spath_prepass(iters)
    full.spath_render(iters)
    ref = full.spath_state()["out_tex"].reshape(h, w, 4)
    full.close()
    y0, th = 16, 16
    tile = rt.Renderer(sc, w, th, tile_y0=y0, full_height=h).build()
    tile.init_spath(seed=seed, resolve_interval=0)
    tile.spath_prepass(iters)
    tile.spath_render(iters)
    got = tile.spath_state()["out_tex"].reshape(th, w, 4)
    tile.close()
    assert_bits_equal(got, ref[y0:y0 + th], "strip vs full frame")
    st = oracle_spath_run(orc, sc, tree, w, th, seed, iters, y0=y0, full_h=h)
    assert_bits_equal(got.reshape(-1, 4), st.out_tex, "strip vs oracle strip")


@pytest.mark.parametrize("h", [48, 56, 61])
def test_interleaved_bands_equal_full_frame(rt, orc, small, h):
    """Rows dealt to 3 ranks in bands of 8: each rank's pixels are the full frame's pixels of those rows.  48 rows
    give every rank two bands; 56 rows are seven bands (shares of 3, 2 and 2 bands); 61 rows end in a band of five
    rows (shares of 24, 21 and 16 rows)."""
    from raytracer_amd.strips import band_partition, band_rows
    sc, tree = small
    w, iters, seed, world = 64, 6, 7, 3
    full = rt.Renderer(sc, w, h).build()
    full.init_spath(seed=seed, resolve_interval=0)
    full.spath_prepass(iters)
    full.spath_render(iters)
    ref = full.spath_state()["out_tex"].reshape(h, w, 4)
    ref_img = full.spath_resolve().reshape(h, w)
    full.close()
    seen = []
    for rank in range(world):
        first, band, stride, rows = band_partition(h, world, rank)
        part = rt.Renderer(sc, w, rows, tile_y0=first, full_height=h, bands=(band, stride)).build()
        part.init_spath(seed=seed, resolve_interval=0)
        part.spath_prepass(iters)
        part.spath_render(iters)
        got = part.spath_state()["out_tex"].reshape(rows, w, 4)
        img = part.spath_resolve().reshape(rows, w)
        part.close()
        where = band_rows(h, world, rank)
        seen += where
        assert_bits_equal(got, ref[where], f"rank {rank} out_tex")
        assert_bits_equal(img, ref_img[where], f"rank {rank} image")
    assert sorted(seen) == list(range(h))


@pytest.mark.parametrize("size", [(256, 256), (300, 250)])
def test_batches_and_heaviest_first_order_do_not_change_results(rt, orc, small, size):
    """A frame large enough for the cost-ordered scheduling (>= 65,536 pixels),
    rendered in three batches (2 + 2 + 1 iterations; the 2nd and 3rd are served
    heaviest first), equals the oracle's iteration-by-iteration result.  256x256
    divides into 8x8 screen tiles and is ordered tile by tile; 300x250 does not
    and is ordered pixel by pixel."""
    sc, tree = small
    (w, h), iters, seed = size, 5, 13
    r = rt.Renderer(sc, w, h).build()
    r.init_spath(seed=seed, resolve_interval=0, max_batch=2)
    r.spath_prepass(num_iterations=iters)
    r.spath_render(iters)
    g = r.spath_state()
    order = r.download(r.spath.contents.cl_pixel_order_buffer, np.uint32, w * h)
    cost = r.download(r.spath.contents.cl_pixel_cost_buffer, np.uint32, w * h)
    tiled = bool(r.spath.contents.cl_tile_scratch_buffer)
    r.close()
    st = oracle_spath_run(orc, sc, tree, w, h, seed, iters)
    assert_bits_equal(g["out_tex"], st.out_tex, "out_tex")
    assert_bits_equal(g["rays"], st.rays, "rays")
    assert_bits_equal(g["progress"], st.progress, "progress")
    assert_bits_equal(g["random"], st.random, "random")
    hits_equal(g["results"], st.results, "hits")
    # the order buffer is a permutation, heaviest bucket first
    assert np.array_equal(np.sort(order), np.arange(w * h, dtype=np.uint32))
    assert cost.max() > 0
    assert tiled == (w % 8 == 0 and h % 8 == 0)
    if tiled:
        # every run of 64 work items is one 8x8 tile, and the tiles come by decreasing cost bucket of their mean
        x, y = (order % w).reshape(-1, 64), (order // w).reshape(-1, 64)
        assert ((x.max(1) - x.min(1)) == 7).all() and ((y.max(1) - y.min(1)) == 7).all()
        assert (x.min(1) % 8 == 0).all() and (y.min(1) % 8 == 0).all()
        mean = cost[order].reshape(-1, 64).sum(1) >> 6
        bucket = np.minimum(mean >> 1, 255)          # the last batch had one iteration: buckets of two steps
        assert (np.diff(bucket.astype(np.int64)) <= 0).all()


@pytest.mark.parametrize("size,max_batch,run_ahead", [((96, 64), None, True), ((256, 256), None, True),
                                                      ((256, 256), 7, True), ((96, 64), None, False)])
def test_frames_at_resolve_points_when_launches_run_ahead(rt, orc, small, size, max_batch, run_ahead):
    """resolve_interval 4, 14 iterations.  Running ahead, the first render call launches all 14 iterations at once
    (or 7 + 7), and the frames the caller is handed at calls 4, 8 and 12 come from the copies of the accumulator
    the kernel kept when each pixel passed those iterations: they must be the oracle's frames after 4, 8 and 12
    iterations, bit for bit, delivered at exactly those calls.  Not running ahead, every launch ends at a resolve
    point and the frames are the same."""
    from raytracer_amd.scenes import splitmix32
    sc, tree = small
    (w, h), iters, seed, every = size, 14, 17, 4
    r = rt.Renderer(sc, w, h).build()
    r.init_spath(seed=seed, resolve_interval=every, max_batch=max_batch, run_ahead=run_ahead)
    r.spath_prepass(num_iterations=iters)
    st = orc.SpathState(sc, tree, w, h)
    st.prepass(splitmix32(seed, np.arange(w * h, dtype=np.uint64)), int(splitmix32(seed + 1, np.arange(1, dtype=np.uint64))[0]))
    rv = splitmix32(seed + 1, np.arange(iters + 1, dtype=np.uint64))
    r.output[:] = 0xdeadbeef
    last = r.output.copy()
    for k in range(1, iters + 1):
        r.L.spath_raytracer_render(r.spath)
        st.iterate(int(rv[k]))
        if k % every == 0:
            assert_bits_equal(r.output, st.resolve(), f"frame handed over at call {k}")
            last = r.output.copy()
        else:
            assert_bits_equal(r.output, last, f"frame untouched at call {k}")
    c = r.spath.contents
    want_launches = {(None, True): 1, (7, True): 2, (None, False): 4}[(max_batch, run_ahead)]
    assert c.launches == want_launches and c.launched_upto == iters
    g = r.spath_state()
    assert_bits_equal(g["out_tex"], st.out_tex, "out_tex after the last iteration")
    assert_bits_equal(g["rays"], st.rays, "rays")
    assert_bits_equal(g["progress"], st.progress, "progress")
    r.L.spath_raytracer_render(r.spath)           # the call after the last iteration: the finished frame
    assert c.render_complete
    assert_bits_equal(r.output, st.resolve(), "finished frame")
    r.close()


@pytest.mark.parametrize("schedule", [0, 1, 2, 3])
def test_deep_stack_scene_bit_exact(rt, orc, schedule):
    """Rays through an 82k-triangle mesh need more stack levels than the LDS part
    holds (the reference's own limit is 16): the spill path must agree too, under
    every slot schedule of the render kernel."""
    sc = rt.scenes.icosphere_scene(6)
    tree = orc.KdTree(sc.verts4, sc.idx4)
    w, h, iters, seed = 128, 96, 4, 21
    r = rt.Renderer(sc, w, h).build()
    r.init_spath(seed=seed, resolve_interval=0, schedule=schedule)
    r.spath_prepass(num_iterations=iters)
    r.spath_render(iters)
    g = r.spath_state()
    r.close()
    st = oracle_spath_run(orc, sc, tree, w, h, seed, iters)
    _, stats = tree.intersect(st.rays, stats=True)
    hits_equal(g["source_results"], st.source_results, "primary hits")
    assert_bits_equal(g["out_tex"], st.out_tex, "out_tex")
    assert_bits_equal(g["rays"], st.rays, "rays")
    assert stats["max_stack"] > 12, stats


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", [0, 1, 2, 3])
def test_long_leaves_bit_exact(rt, orc, schedule):
    """Leaves of 45 and 33 triangles (stacks of coincident ones): more than a leaf descriptor's count field holds, so
    the walk reads the count from the leaf's header slot; of a stack the first triangle in stored order is reported,
    as by the reference's strict t < t_hit."""
    sc = rt.scenes.bundle_scene()
    tree = orc.KdTree(sc.verts4, sc.idx4)
    w, h, iters, seed = 96, 64, 6, 5
    r = rt.Renderer(sc, w, h).build()
    r.init_spath(seed=seed, resolve_interval=0, schedule=schedule)
    r.spath_prepass(num_iterations=iters)
    r.spath_render(iters)
    g = r.spath_state()
    r.close()
    st = oracle_spath_run(orc, sc, tree, w, h, seed, iters)
    hits_equal(g["source_results"], st.source_results, "primary hits")
    hits_equal(g["results"], st.results, "bounce hits")
    assert_bits_equal(g["out_tex"], st.out_tex, "out_tex")
    assert_bits_equal(g["rays"], st.rays, "rays")
    first_of_stack = np.isin(st.source_results["tri"], [0, 3 * 45]) & (st.source_results["t"] != 0)
    in_stack = (st.source_results["tri"] < 3 * (45 + 33)) & (st.source_results["t"] != 0)
    assert first_of_stack.sum() > 100 and (first_of_stack == in_stack).all()


def test_scene_file_to_pixels(rt, orc, tmp_path):
    """The whole drop-in chain: .rsc + OBJ files -> load_scene_json_url -> build -> SPATH, against the oracle
    fed with the arrays the loader produced."""
    url = rt.scenes.write_rsc(rt.scenes.icosphere_scene(3), str(tmp_path), "ico3")
    sc = rt.load_scene(url, str(tmp_path))
    assert sc.num_tris == 20 * 4 ** 3 + 6 and len(sc.meshes) == 4
    w, h, iters, seed = 80, 48, 7, 4
    r = rt.Renderer(sc, w, h).build()
    r.init_spath(seed=seed, resolve_interval=0)
    r.spath_prepass(num_iterations=iters)
    r.spath_render(iters)
    tree = orc.KdTree(sc.verts4, sc.idx4)
    st = oracle_spath_run(orc, sc, tree, w, h, seed, iters)
    g = r.spath_state()
    hits_equal(g["source_results"], st.source_results, "primary hits")
    hits_equal(g["results"], st.results, "bounce hits")
    assert_bits_equal(g["out_tex"], st.out_tex, "out_tex")
    assert_bits_equal(r.spath_resolve(), st.resolve(), "resolved ARGB image")
    assert (st.source_results["t"] != 0).sum() > 500
    r.close()


def test_headless_front_end(rt, orc, tmp_path):
    """rt_headless (C, the reference's render-thread call sequence) renders a scene file to a PPM; the pixels are
    the resolved frame of the same run through the binding."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(rt.LIB_PATH), "rt_headless")
    assert os.path.exists(exe), "rt_headless is built by raytracer_amd/csrc/Makefile"
    url = rt.scenes.write_rsc(rt.scenes.icosphere_scene(3), str(tmp_path), "ico3")
    w, h, samples = 64, 48, 3
    out = tmp_path / "frame.ppm"
    env = dict(os.environ, RT_RNG_SEED="9")
    p = subprocess.run([exe, str(tmp_path), url, str(w), str(h), str(samples), str(out), "spath"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 0, p.stderr.decode()[-1500:] + p.stdout.decode()[-1500:]
    raw = out.read_bytes()
    header = f"P6\n{w} {h}\n255\n".encode()
    assert raw.startswith(header) and len(raw) == len(header) + 3 * w * h
    rgb = np.frombuffer(raw[len(header):], np.uint8).reshape(h * w, 3).astype(np.uint32)
    assert len(np.unique(rgb, axis=0)) > 20          # an image, not a constant
    # same scene, seed and iteration count through the binding and through the oracle
    sc = rt.load_scene(url, str(tmp_path))
    r = rt.Renderer(sc, w, h).build()
    r.init_spath(seed=9, resolve_interval=50)
    r.spath_prepass(num_iterations=samples * 4)
    r.spath_render(samples * 4 + 1)                  # the call after the last iteration resolves the frame
    frame = r.output.copy()
    r.close()
    assert_bits_equal((rgb[:, 0] << 16) | (rgb[:, 1] << 8) | rgb[:, 2], frame & 0xffffff, "PPM pixels")
    st = oracle_spath_run(orc, sc, orc.KdTree(sc.verts4, sc.idx4), w, h, 9, samples * 4)
    assert_bits_equal(frame, st.resolve(), "frame against the oracle")


@pytest.mark.parametrize("size", [(1, 1), (33, 17), (65, 3), (7, 130)])
def test_odd_frame_sizes(rt, orc, small, size):
    """Frames that fill no wave, a fraction of one, or end in a ragged one."""
    sc, tree = small
    w, h = size
    iters, seed = 6, 21
    r = rt.Renderer(sc, w, h).build()
    r.init_spath(seed=seed, resolve_interval=0)
    r.spath_prepass(num_iterations=iters)
    r.spath_render(iters)
    st = oracle_spath_run(orc, sc, tree, w, h, seed, iters)
    g = r.spath_state()
    hits_equal(g["source_results"], st.source_results, "primary hits")
    hits_equal(g["results"], st.results, "bounce hits")
    assert_bits_equal(g["rays"], st.rays, "rays")
    assert_bits_equal(g["progress"], st.progress, "progress")
    assert_bits_equal(g["out_tex"], st.out_tex, "out_tex")
    assert_bits_equal(r.spath_resolve(), st.resolve(), "image")
    r.close()


def test_single_leaf_tree(rt, orc):
    """Two triangles: the builder refuses to split, the root is a leaf.  The reference walks such a tree as if the
    root were an interior node (undefined); here and in the oracle the leaf is tested once."""
    S = rt.scenes
    sc = S._assemble([S._quad((-2, -2, -5), (2, -2, -6), (2, 2, -5), (-2, 2, -6), (0, 0, 1)) + (3,)],
                     S._materials(S._PATH_MATERIALS))
    tree = orc.KdTree(sc.verts4, sc.idx4)
    w, h, iters, seed = 40, 24, 5, 2
    r = rt.Renderer(sc, w, h).build()
    r.init_spath(seed=seed, resolve_interval=0)
    r.spath_prepass(num_iterations=iters)
    assert r.kdt.view.root_ref & 0x80000000
    r.spath_render(iters)
    st = oracle_spath_run(orc, sc, tree, w, h, seed, iters)
    g = r.spath_state()
    hits_equal(g["source_results"], st.source_results, "primary hits")
    assert 0 < (st.source_results["t"] != 0).sum() < w * h
    hits_equal(g["results"], st.results, "bounce hits")
    assert_bits_equal(g["out_tex"], st.out_tex, "out_tex")
    r.close()


def test_kd_intersect_arbitrary_rays(rt, orc, small):
    """rt_k_kd_intersect on rays no camera produces: origins inside and far outside the scene box, directions with
    exact zeros, negative zeros, denormal and huge components, all-zero rays; a count that fills no whole wave."""
    import ctypes as C
    sc, tree = small
    rng = np.random.RandomState(5)
    n = 10007
    o = rng.uniform(-12, 12, size=(n, 3)).astype(np.float32)
    o[: n // 3] = rng.uniform(-40, 40, size=(n // 3, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    special = np.array([0.0, -0.0, 1e-42, -1e-42, 1e30, -1e30, 1.0, -1.0], np.float32)
    for k in range(0, 3000):                 # sprinkle special values into single components
        d[k, rng.randint(3)] = special[rng.randint(len(special))]
    d[3000:3010] = 0.0
    o[3010:3020] = 0.0
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3], rays[:, 4:7] = o, d
    r = rt.Renderer(sc, 8, 8).build()
    r.L.raytracer_prepass(r.rctx)            # uploads the flattened tree
    ro = r.L.rt_mem_alloc(r.dev, 16 * n); rd = r.L.rt_mem_alloc(r.dev, 16 * n); hits = r.L.rt_mem_alloc(r.dev, 16 * n)
    r.upload(ro, np.ascontiguousarray(rays[:, 0:4])); r.upload(rd, np.ascontiguousarray(rays[:, 4:8]))
    r.L.rt_k_kd_intersect(r.dev, C.byref(r.kdt.view), ro, rd, hits, n)
    r.sync()
    got = r.download(hits, np.uint32, 4 * n).reshape(n, 4)
    want = tree.intersect(rays)
    assert_bits_equal(got[:, 0], want["tri"], "index offset")
    for k, f in ((1, "t"), (2, "u"), (3, "v")):
        assert_bits_equal(got[:, k], want[f].view(np.uint32), f)
    assert 0.05 < (want["t"] != 0).mean() < 0.95
    for m in (ro, rd, hits):
        r.L.rt_mem_free(r.dev, m)
    r.close()


def test_default_rng_mode_renders_the_same_picture(rt, tmp_path):
    """Without a seed the library draws from srand(clock) / rand() like the reference: not reproducible, but the
    converged picture must be the same as the seeded one up to noise."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(rt.LIB_PATH), "rt_headless")
    url = rt.scenes.write_rsc(rt.scenes.icosphere_scene(3), str(tmp_path), "ico3")
    w, h, samples = 64, 48, 24
    frames = []
    for name, extra in (("seeded", {"RT_RNG_SEED": "3"}), ("libc", {})):
        out = tmp_path / f"{name}.ppm"
        env = {k: v for k, v in os.environ.items() if k != "RT_RNG_SEED"}
        env.update(extra)
        p = subprocess.run([exe, str(tmp_path), url, str(w), str(h), str(samples), str(out), "spath"], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert p.returncode == 0, p.stderr.decode()[-1500:]
        raw = out.read_bytes()
        frames.append(np.frombuffer(raw[len(f"P6\n{w} {h}\n255\n"):], np.uint8).astype(np.float64))
    a, b = frames
    assert not np.array_equal(a, b)                       # different random streams
    assert abs(a.mean() - b.mean()) < 0.05 * a.mean()     # same picture
    assert np.sqrt(np.mean((a - b) ** 2)) < 40            # per-pixel noise of 24 samples, not a different image


def test_primary_visibility_matches_the_reference_stored_render(rt, orc):
    """The library's prepass at 1920x1080 on the reference's path_obj2 scene against the frame of it that the
    reference stores (emitter silhouette pixel for pixel, see tests/golden/make_ref_render_fixture.py), and bit
    for bit against the oracle."""
    from helpers import load_ref_render_fixture, check_against_stored_render
    sc, w, h, emitter, objects = load_ref_render_fixture(rt)
    r = rt.Renderer(sc, w, h).build()
    r.init_spath(seed=1, resolve_interval=0)
    r.spath_prepass(num_iterations=4)
    g = r.spath_state()
    r.close()
    check_against_stored_render(sc, g["source_results"], emitter, objects)
    st = oracle_spath_run(orc, sc, orc.KdTree(sc.verts4, sc.idx4), w, h, 1, 0)
    hits_equal(g["source_results"], st.source_results, "primary hits")


def test_traversal_against_frozen_outputs_of_the_reference_kernel(rt):
    """tests/golden/refkernel_ico4.npz: 22,464 rays and the hit records the REFERENCE's own kdtree_intersection
    (compiled unmodified for gfx950, run on an MI355X; make_refkernel_fixture.py) returned for them.  The library's
    traversal, through the C API, returns the same records bit for bit.  Needs nothing of the reference at run time."""
    import ctypes as C
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refkernel_ico4.npz"))
    rays, want = np.ascontiguousarray(z["rays"]), z["hits"]
    n = rays.shape[0]
    sc = rt.scenes.icosphere_scene(4)
    r = rt.Renderer(sc, 8, 8).build()
    r.L.raytracer_prepass(r.rctx)
    ro = r.L.rt_mem_alloc(r.dev, 16 * n); rd = r.L.rt_mem_alloc(r.dev, 16 * n); hits = r.L.rt_mem_alloc(r.dev, 16 * n)
    r.upload(ro, np.ascontiguousarray(rays[:, 0:4])); r.upload(rd, np.ascontiguousarray(rays[:, 4:8]))
    r.L.rt_k_kd_intersect(r.dev, C.byref(r.kdt.view), ro, rd, hits, n)
    r.sync()
    got = r.download(hits, np.uint32, 4 * n).reshape(n, 4)
    assert_bits_equal(got[:, 0], want["tri"], "index offset")
    for k, f in ((1, "t"), (2, "u"), (3, "v")):
        assert_bits_equal(got[:, k], want[f].view(np.uint32), f)
    for m in (ro, rd, hits):
        r.L.rt_mem_free(r.dev, m)
    r.close()


def test_second_prepass_starts_a_fresh_render(rt, orc, small):
    """A context that has rendered (launches that ran ahead, snapshots taken, pixel order learnt) and is given a new
    prepass renders exactly what a fresh context renders: nothing of the first render's launch state survives."""
    sc, tree = small
    w, h, seed, every = 256, 256, 23, 4
    r = rt.Renderer(sc, w, h).build()
    r.init_spath(seed=seed, resolve_interval=every)
    r.spath_prepass(num_iterations=11)
    r.spath_render(11)
    first = r.spath_state()["out_tex"].copy()
    r.spath_prepass(num_iterations=6)             # restarts the random streams of the seeded mode as well
    frames = []
    for k in range(1, 7):
        r.L.spath_raytracer_render(r.spath)
        if k % every == 0:
            frames.append(r.output.copy())
    again = r.spath_state()
    c = r.spath.contents
    assert c.launches == 1 and c.launched_upto == 6 and c.current_iteration == 6
    r.close()
    st = oracle_spath_run(orc, sc, tree, w, h, seed, 4)
    assert_bits_equal(frames[0], st.resolve(), "frame at call 4 of the second render")
    for k in range(2):
        st.iterate(int(rt.scenes.splitmix32(seed + 1, np.arange(7, dtype=np.uint64))[5 + k]))
    assert_bits_equal(again["out_tex"], st.out_tex, "out_tex of the second render")
    assert_bits_equal(again["progress"], st.progress, "progress of the second render")
    assert not np.array_equal(first, again["out_tex"])


def test_checkpoint_and_resume(rt, orc, small, tmp_path):
    """10 iterations, save, a new context on the same scene, restore, 6 more: every per-pixel buffer equals the
    uninterrupted 16-iteration render's (the oracle's), bit for bit; a checkpoint of another frame size is refused."""
    sc, tree = small
    w, h, seed = 96, 64, 29
    path = str(tmp_path / "render.ckpt").encode()
    a = rt.Renderer(sc, w, h).build()
    a.init_spath(seed=seed, resolve_interval=0)
    a.spath_prepass(num_iterations=16)
    a.spath_render(10)
    assert a.L.spath_raytracer_save(a.spath, path) == 0
    a.close()
    b = rt.Renderer(sc, w, h).build()
    b.init_spath(seed=seed + 100, resolve_interval=0)        # the checkpoint brings its own seed and stream position
    b.spath_prepass(num_iterations=3)
    assert b.L.spath_raytracer_restore(b.spath, path) == 0
    assert b.spath.contents.current_iteration == 10 and b.spath.contents.num_iterations == 16
    b.spath_render(6)
    g = b.spath_state()
    img = b.spath_resolve()
    b.close()
    st = oracle_spath_run(orc, sc, tree, w, h, seed, 16)
    hits_equal(g["results"], st.results, "bounce hits")
    for name in ("rays", "origin_rays", "progress", "random", "out_tex"):
        assert_bits_equal(g[name], getattr(st, name), name)
    assert_bits_equal(img, st.resolve(), "resolved frame")
    c = rt.Renderer(sc, w, h + 8).build()
    c.init_spath(seed=seed, resolve_interval=0)
    c.spath_prepass(num_iterations=3)
    assert c.L.spath_raytracer_restore(c.spath, path) == 2
    assert c.L.spath_raytracer_restore(c.spath, b"/nonexistent/file") == 1
    assert c.spath.contents.current_iteration == 0
    c.close()
    # damaged headers are refused (2) and leave the context alone: an iteration past the render's end, an RNG mode
    # that does not exist, another layout version.  Header: magic[8], version, header_bytes, width, height, rows[4],
    # num_tris, tree_bytes, iteration, num_iterations, rng_mode, ...
    raw = bytearray(open(path.decode(), "rb").read())
    words = np.frombuffer(bytes(raw[:96]), np.uint32).copy()
    assert words[2] == 2 and words[3] == 84 and words[12] == 10 and words[13] == 16 and words[14] == rt.RT_RNG_SPLITMIX
    d = rt.Renderer(sc, w, h).build()
    d.init_spath(seed=seed, resolve_interval=0)
    d.spath_prepass(num_iterations=3)
    for index, value in ((12, 17), (14, 7), (2, 1), (3, 80)):
        bad = words.copy()
        bad[index] = value
        damaged = str(tmp_path / f"damaged_{index}.ckpt")
        with open(damaged, "wb") as f:
            f.write(bad[:21].tobytes() + bytes(raw[84:]))
        assert d.L.spath_raytracer_restore(d.spath, damaged.encode()) == 2, index
        assert d.spath.contents.current_iteration == 0 and d.spath.contents.num_iterations == 3
    d.close()


def test_changing_the_plan_while_launches_are_ahead_is_refused(rt):
    """Through the C API, as a reference-style caller would: with run_ahead the first render call launches the whole
    render; lowering num_iterations below what is launched, switching to separate kernels, or asking for the frame
    between resolve points would then hand over frames with iterations nobody asked for -- the library refuses
    (message, exit status 1, the host API's error convention).  With run_ahead = 0 the same calls are fine."""
    import subprocess
    import sys
    prog = """
import sys
sys.path.insert(0, {root!r})
import raytracer_amd as rt
sc = rt.scenes.icosphere_scene(3)
r = rt.Renderer(sc, 64, 64).build()
r.init_spath(seed=3, resolve_interval=50, run_ahead={ahead})
r.L.spath_raytracer_prepass(r.spath)
c = r.spath.contents
c.num_iterations = 120
r.L.spath_raytracer_render(r.spath)
print("launched", c.launched_upto, flush=True)
what = {what!r}
if what == "lower":
    c.num_iterations = 10
    r.L.spath_raytracer_render(r.spath)
elif what == "separate":
    c.separate_kernels = 1
    r.L.spath_raytracer_render(r.spath)
elif what == "resolve":
    r.L.spath_raytracer_resolve(r.spath)
r.sync()
print("survived", c.launched_upto, flush=True)
"""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for what in ("lower", "separate", "resolve"):
        for ahead in (1, 0):
            p = subprocess.run([sys.executable, "-c", prog.format(root=root, ahead=ahead, what=what)],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
            out, err = p.stdout.decode(), p.stderr.decode()
            if ahead:
                assert "launched 120" in out and "survived" not in out, (what, out, err[-500:])
                assert p.returncode == 1 and "already launched" in err, (what, p.returncode, err[-500:])
            else:
                # nothing is launched beyond what was asked for (iterations are collected, or launched one by one)
                assert p.returncode == 0 and "launched 120" not in out and "survived" in out, (what, out, err[-500:])


def test_a_talking_context_leaves_the_device_timing_table_alone(rt, small, capfd):
    """A context that prints the line at its resolve points (quiet = false) times its own launches with markers of
    its own: the device-wide rt_timing_* table keeps what another user of the rt_device has collected in it."""
    sc, _ = small
    r = rt.Renderer(sc, 64, 64).build()
    r.init_spath(seed=5, resolve_interval=50, quiet=False)
    r.timing(True); r.timing_reset()
    r.spath_prepass(num_iterations=120)
    before, _ = r.timing_get("kd_intersect")
    assert before == 1                                   # the prepass's traversal launch, counted by the table
    r.spath_render(110)
    r.sync()
    launches, ms = r.timing_get("spath_run")
    assert launches >= 1 and ms > 0                      # not cleared at the resolve points at iterations 50 and 100
    assert r.timing_get("kd_intersect")[0] == 1
    out = capfd.readouterr().out
    assert out.count("collision+trace:") == 2 and "launches)" in out
    r.timing(False)
    r.close()


def test_headless_event_protocol(rt, orc, tmp_path):
    """rt_headless in script mode replays the reference's UI messages and render-thread loop (src/ui.c:24-98,
    src/startup.c:118-216): events go through rctx->event_stack last in first out, each switch runs the integrator's
    prepass, every turn renders once with the current integrator, {"type":2} hands out the serialised kd-tree."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(rt.LIB_PATH), "rt_headless")
    url = rt.scenes.write_rsc(rt.scenes.icosphere_scene(3), str(tmp_path), "ico3")
    w, h, samples = 64, 48, 3
    out = tmp_path / "final.ppm"
    script = tmp_path / "events.txt"
    script.write_text("\n".join([
        '{"type":0}',
        '{"type":1,"action":{"type":0}}', "turn 1", f"frame {tmp_path / 'ss.ppm'}",
        '{"type":1,"action":{"type":2}}', '{"type":1,"action":{"type":1}}',      # path is queued last: popped first
        "turn 1", "turn 1", f"turn {samples * 4}", f"frame {tmp_path / 'spath.ppm'}",
        '{"type": 2}', '{"type":1,"action":{"type":10}}', "turn 2", ""]))
    env = dict(os.environ, RT_RNG_SEED="9")
    p = subprocess.run([exe, str(tmp_path), url, str(w), str(h), str(samples), str(out), "script", str(script)], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 0, p.stderr.decode()[-1500:] + p.stdout.decode()[-1500:]
    log = p.stdout.decode()
    order = [log.index(s) for s in ("Nothing Right Now", "Switching To SS Raytracer", "Switching To Path Tracer",
                                    "Switching To Split Path Tracer", "GE2 requested k-d tree", "Clearing")]
    assert order == sorted(order), log

    def pixels(path):
        raw = open(path, "rb").read()
        header = f"P6\n{w} {h}\n255\n".encode()
        rgb = np.frombuffer(raw[len(header):], np.uint8).reshape(h * w, 3).astype(np.uint32)
        return (rgb[:, 0] << 16) | (rgb[:, 1] << 8) | rgb[:, 2]

    sc = rt.load_scene(url, str(tmp_path))
    r = rt.Renderer(sc, w, h).build()
    assert_bits_equal(pixels(tmp_path / "ss.ppm"), r.ss_render() & 0xffffff, "single-sweep frame")
    r.init_spath(seed=9, resolve_interval=50)
    r.spath_prepass(num_iterations=samples * 4)
    r.spath_render(samples * 4 + 1)
    assert_bits_equal(pixels(tmp_path / "spath.ppm"), r.output & 0xffffff, "split path tracer frame")
    blob = np.ctypeslib.as_array(r.kdt.buffer, shape=(int(r.kdt.buffer_size),)).copy()
    r.close()
    assert_bits_equal(np.fromfile(str(out) + ".kdtree", np.uint8), blob, "serialised kd-tree")
    assert (pixels(out) == 0x404040).all()          # the clear event's placeholder (os_draw_weird in the reference)
