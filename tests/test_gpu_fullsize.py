"""GPU parity at the benchmark's full size (1.31 M triangles, 2048x2048): the oracle checks what it can finish in
seconds (a strip of the frame, a sample of the bounce rays), size-independent properties cover the rest (the batched
kernel against one launch per kernel per iteration over the whole frame)."""
import ctypes as C

import numpy as np
import pytest

from helpers import assert_bits_equal, hits_equal, oracle_spath_run

pytestmark = pytest.mark.gpu

W = H = 2048
SEED = 11


@pytest.fixture(scope="module")
def c3(rt):
    return rt.scenes.icosphere_scene(8)


@pytest.fixture(scope="module")
def batched(rt, c3):
    """Full frame, batched kernel with pixel reordering, 7 iterations in batches of 3 + 3 + 1."""
    r = rt.Renderer(c3, W, H).build()
    r.init_spath(seed=SEED, resolve_interval=0, quiet=True, max_batch=3)
    r.spath_prepass(num_iterations=1 << 30)
    r.spath_render(7)
    yield r
    r.close()


def _oracle_tree(orc, r, sc):
    kdt = r.kdt
    blob = np.ctypeslib.as_array(kdt.buffer, shape=(int(kdt.buffer_size),))
    return orc.KdTree.from_blob(blob, sc.verts4, sc.idx4, kdt.bounds.min[:3], kdt.bounds.max[:3])


def test_batched_equals_per_iteration_launches(rt, c3, batched):
    """Two independent code paths over all 4.19 M pixels: k_spath_run (lanes keep a pixel for a batch, heaviest
    pixels first) against xorshift + kd_intersect + spath_trace launched one by one."""
    ref = rt.Renderer(c3, W, H).build()
    ref.init_spath(seed=SEED, separate_kernels=True, resolve_interval=0, quiet=True)
    ref.spath_prepass(num_iterations=1 << 30)
    ref.spath_render(7)
    n = W * H
    a, b = batched.spath_buffers(), ref.spath_buffers()
    for name, words in (("ray_o", 4), ("ray_d", 4), ("hits", 4), ("prog_a", 4), ("prog_b", 4), ("random", 1),
                        ("out_tex", 4)):
        ga = batched.download(getattr(a, name), np.uint32, words * n)
        gb = ref.download(getattr(b, name), np.uint32, words * n)
        assert_bits_equal(ga, gb, name)
    assert_bits_equal(batched.spath_resolve(), ref.spath_resolve(), "resolved image")
    ref.close()


def test_bounce_hits_against_oracle_sample(rt, orc, c3, batched):
    """The rays the frame holds after 7 iterations (incoherent bounce rays), traced by k_kd_intersect; every 23rd
    ray checked against the oracle's walk of the same serialised tree."""
    r = batched
    n = W * H
    b = r.spath_buffers()
    scratch = r.L.rt_mem_alloc(r.dev, 16 * n)
    r.L.rt_k_kd_intersect(r.dev, C.byref(r.kdt.view), b.ray_o, b.ray_d, scratch, n)
    r.sync()
    hits = r.download(scratch, np.uint32, 4 * n).reshape(n, 4)
    r.L.rt_mem_free(r.dev, scratch)
    ro = r.download(b.ray_o, np.float32, 4 * n).reshape(n, 4)
    rd = r.download(b.ray_d, np.float32, 4 * n).reshape(n, 4)
    pick = np.arange(0, n, 23)
    rays = np.ascontiguousarray(np.concatenate([ro[pick], rd[pick]], axis=1))
    tree = _oracle_tree(orc, r, c3)
    want = tree.intersect(rays)
    got = hits[pick]
    assert_bits_equal(got[:, 0], want["tri"], "hit index offset")
    for k, f in ((1, "t"), (2, "u"), (3, "v")):
        assert_bits_equal(got[:, k], want[f].view(np.uint32), f"hit {f}")
    assert (want["t"] != 0).mean() > 0.2          # a real mix of hits and misses


def test_strip_of_the_full_frame_against_oracle(rt, orc, c3):
    """Rows 1020..1027 of the 2048-row frame (through the sphere), 12 iterations: every per-pixel buffer bit for bit."""
    y0, rows, iters = 1020, 8, 12
    r = rt.Renderer(c3, W, rows, tile_y0=y0, full_height=H).build()
    r.init_spath(seed=SEED, resolve_interval=0, quiet=True)
    r.spath_prepass(num_iterations=1 << 30)
    r.spath_render(iters)
    tree = _oracle_tree(orc, r, c3)
    st = oracle_spath_run(orc, c3, tree, W, rows, SEED, iters, y0=y0, full_h=H)
    g = r.spath_state()
    hits_equal(g["source_results"], st.source_results, "primary hits")
    hits_equal(g["results"], st.results, "bounce hits")
    assert_bits_equal(g["rays"], st.rays, "rays")
    assert_bits_equal(g["progress"], st.progress, "progress")
    assert_bits_equal(g["random"], st.random, "random")
    assert_bits_equal(g["out_tex"], st.out_tex, "out_tex")
    assert_bits_equal(r.spath_resolve(), st.resolve(), "resolved ARGB strip")
    assert (st.source_results["t"] != 0).mean() > 0.5
    r.close()


def test_soup_scene_strip_against_oracle(rt, orc):
    """Unstructured geometry (random triangle soup, 1 M triangles: long leaves, deep stacks)."""
    sc = rt.scenes.soup_scene(1_000_000)
    w, h_full, y0, rows, iters = 512, 512, 250, 6, 8
    r = rt.Renderer(sc, w, rows, tile_y0=y0, full_height=h_full).build()
    r.init_spath(seed=3, resolve_interval=0, quiet=True)
    r.spath_prepass(num_iterations=1 << 30)
    r.spath_render(iters)
    tree = _oracle_tree(orc, r, sc)
    st = oracle_spath_run(orc, sc, tree, w, rows, 3, iters, y0=y0, full_h=h_full)
    g = r.spath_state()
    hits_equal(g["results"], st.results, "bounce hits")
    assert_bits_equal(g["progress"], st.progress, "progress")
    assert_bits_equal(g["out_tex"], st.out_tex, "out_tex")
    r.close()


def test_config2_full_frame_both_launch_structures(rt):
    """82 k-triangle scene at 1024x1024 (the ~70 k-triangle configuration): 64 iterations batched (with the
    schedule calibration, pixel reordering and run-ahead launches on their defaults) against one launch per kernel per iteration."""
    sc = rt.scenes.icosphere_scene(6)
    w = h = 1024
    a = rt.Renderer(sc, w, h).build()
    a.init_spath(seed=5, resolve_interval=50, quiet=True)
    a.spath_prepass(num_iterations=1 << 30)
    a.spath_render(64)
    b = rt.Renderer(sc, w, h).build()
    b.init_spath(seed=5, separate_kernels=True, resolve_interval=50, quiet=True)
    b.spath_prepass(num_iterations=1 << 30)
    b.spath_render(64)
    n = w * h
    ba, bb = a.spath_buffers(), b.spath_buffers()
    for name, words in (("ray_o", 4), ("ray_d", 4), ("hits", 4), ("prog_a", 4), ("prog_b", 4), ("random", 1),
                        ("out_tex", 4)):
        assert_bits_equal(a.download(getattr(ba, name), np.uint32, words * n),
                          b.download(getattr(bb, name), np.uint32, words * n), name)
    # the frame the caller saw at the last resolve point (iteration 50) is the same too: `a` ran all 64 iterations
    # in one launch and resolved the copy of the accumulator it kept at iteration 50
    assert_bits_equal(a.output, b.output, "frame at the resolve point")
    assert a.spath.contents.trial_state == 2 and a.spath.contents.launches == 1
    assert min(a.spath.contents.trial_ms) > 0
    # launches that hold only what has been asked for end at the resolve point: 50 + 14
    c = rt.Renderer(sc, w, h).build()
    c.init_spath(seed=5, resolve_interval=50, quiet=True, run_ahead=False)
    c.spath_prepass(num_iterations=1 << 30)
    c.spath_render(64)
    assert c.spath.contents.launches == 2
    assert_bits_equal(c.output, b.output, "frame at the resolve point, launches not running ahead")
    bc = c.spath_buffers()
    assert_bits_equal(c.download(bc.out_tex, np.uint32, 4 * n), b.download(bb.out_tex, np.uint32, 4 * n), "out_tex")
    a.close(); b.close(); c.close()


def test_config5_size_strip_against_oracle(rt, orc):
    """10 M-triangle soup, four rows through the middle of a 3840x2160 frame: tree flattening at that size
    (11 M nodes, 30 M leaf slots), long leaves, every slot schedule, against the oracle."""
    sc = rt.scenes.soup_scene(10_000_000)
    w, full_h, y0, rows, iters = 3840, 2160, 1078, 4, 5
    for schedule in (1, 0, 2, 3):
        r = rt.Renderer(sc, w, rows, tile_y0=y0, full_height=full_h).build()
        r.init_spath(seed=8, resolve_interval=0, quiet=True, schedule=schedule)
        r.spath_prepass(num_iterations=1 << 30)
        r.spath_render(iters)
        if schedule == 1:
            tree = _oracle_tree(orc, r, sc)
            st = oracle_spath_run(orc, sc, tree, w, rows, 8, iters, y0=y0, full_h=full_h)
            del tree
        g = r.spath_state()
        hits_equal(g["source_results"], st.source_results, "primary hits")
        hits_equal(g["results"], st.results, "bounce hits")
        assert_bits_equal(g["progress"], st.progress, "progress")
        assert_bits_equal(g["out_tex"], st.out_tex, "out_tex")
        r.close()
    assert (st.source_results["t"] != 0).mean() > 0.3


def _strip_against_oracle(rt, orc, sc, w, h, y0, rows, iters, seed):
    r = rt.Renderer(sc, w, rows, tile_y0=y0, full_height=h).build()
    r.init_spath(seed=seed, resolve_interval=0, quiet=True)
    r.spath_prepass(num_iterations=iters)
    r.spath_render(iters)
    tree = _oracle_tree(orc, r, sc)
    st = oracle_spath_run(orc, sc, tree, w, rows, seed, iters, y0=y0, full_h=h)
    g = r.spath_state()
    hits_equal(g["source_results"], st.source_results, "primary hits")
    hits_equal(g["results"], st.results, "bounce hits")
    assert_bits_equal(g["rays"], st.rays, "rays")
    assert_bits_equal(g["progress"], st.progress, "progress")
    assert_bits_equal(g["random"], st.random, "random")
    assert_bits_equal(g["out_tex"], st.out_tex, "out_tex")
    assert_bits_equal(r.spath_resolve(), st.resolve(), "resolved ARGB strip")
    r.close()
    return st


def test_config2_on_the_reference_geometry(rt, orc):
    """Configuration 2 on the reference's own files: res/scenes/path_obj3.rsc with mesh/test3.obj (its largest mesh,
    46,966 triangles) where the scene names the missing happy.obj, light.obj and indigo_happy_walls.obj as the scene
    arranges them (tests/golden/c2obj_scene.npz, written by make_c2obj_fixture.py from the reference's files through
    the library's loader).  1024x1024: eight rows through the mesh against the oracle, bit for bit, over 13
    iterations (three finished samples); then the whole frame, 64 iterations in one launch against one launch per
    kernel per iteration, including the frame handed over at iteration 50."""
    import os
    sc = rt.scenes.scene_from_npz(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c2obj_scene.npz"))
    assert sc.num_tris == 46972
    w = h = 1024
    st = _strip_against_oracle(rt, orc, sc, w, h, y0=448, rows=8, iters=13, seed=6)
    mesh_of_hit = sc.idx4.reshape(-1, 4)[st.source_results["tri"], 3][st.source_results["t"] != 0]
    assert (mesh_of_hit == 0).sum() > 200            # the strip crosses test3.obj itself, not only the walls
    a = rt.Renderer(sc, w, h).build()
    a.init_spath(seed=6, resolve_interval=50, quiet=True)
    a.spath_prepass(num_iterations=64)
    a.spath_render(64)
    b = rt.Renderer(sc, w, h).build()
    b.init_spath(seed=6, separate_kernels=True, resolve_interval=50, quiet=True)
    b.spath_prepass(num_iterations=64)
    b.spath_render(64)
    n = w * h
    ba, bb = a.spath_buffers(), b.spath_buffers()
    for name, words in (("ray_o", 4), ("ray_d", 4), ("hits", 4), ("prog_a", 4), ("prog_b", 4), ("random", 1),
                        ("out_tex", 4)):
        assert_bits_equal(a.download(getattr(ba, name), np.uint32, words * n),
                          b.download(getattr(bb, name), np.uint32, words * n), name)
    assert_bits_equal(a.output, b.output, "frame at the resolve point")
    assert a.spath.contents.launches == 1
    a.close(); b.close()


def test_reference_scene_path_obj2_bounces_against_oracle(rt, orc):
    """The reference's scene res/scenes/path_obj2.rsc (the fixture of its stored render) beyond primary visibility:
    eight rows of the 1920x1080 frame through the objects, 9 iterations of bounces, every buffer bit for bit."""
    from helpers import load_ref_render_fixture
    sc, w, h, _, _ = load_ref_render_fixture(rt)
    st = _strip_against_oracle(rt, orc, sc, w, h, y0=600, rows=8, iters=9, seed=4)
    assert (st.source_results["t"] != 0).mean() > 0.5 and st.out_tex[:, 3].max() >= 2
