"""GPU parity for the brute-force integrators (SS ray caster, PATH tracer)."""
import numpy as np
import pytest

from helpers import assert_bits_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene_name", ["sphere_plane", "mixed"])
def test_ss_cast_rays_bit_exact(rt, orc, scene_name):
    sc = rt.scenes.sphere_plane_scene() if scene_name == "sphere_plane" else rt.scenes.mixed_scene(2)
    w, h = (128, 128) if scene_name == "sphere_plane" else (96, 64)
    r = rt.Renderer(sc, w, h).build()
    img = r.ss_render()
    want = orc.cast_ray_test(sc, orc.generate_rays(w, h))
    assert_bits_equal(img, want, "cast_ray_test image")
    assert len(np.unique(img)) > 20
    r.close()


def test_path_trace_within_tolerance(rt, orc):
    """path_trace seeds its RNG through sin() of large arguments, so device and
    host libm may disagree on a few pixels' seeds; everything else is exact.
    Tolerance: at most 0.5% of pixels differ, and the image RMSE is below 2e-2."""
    sc = rt.scenes.mixed_scene(2)
    w, h = 64, 48
    r = rt.Renderer(sc, w, h).build()
    r.init_path(num_samples=3)
    rays = orc.generate_rays(w, h)
    out = None
    for sample in (1, 2, 3):
        img = r.path_render()
        fresh = orc.path_trace(sc, rays, w, h, (0, 0, 0), sample)
        if sample == 1:
            out = fresh.copy()
        orc.f_buffer_average(out, fresh, sample)
        got_fresh = r.download(r.path.contents.cl_path_fresh_frame_buffer, np.float32, 4 * w * h).reshape(-1, 4)
        differ = (got_fresh.view(np.uint32) != fresh.view(np.uint32)).any(axis=1).mean()
        assert differ <= 0.005, f"{differ:.4f} of pixels differ in sample {sample}"
        rmse = float(np.sqrt(np.mean((got_fresh - fresh) ** 2)))
        assert rmse < 2e-2
        got_out = r.download(r.path.contents.cl_path_output_buffer, np.float32, 4 * w * h).reshape(-1, 4)
        if differ == 0:
            assert_bits_equal(got_out, out, "running mean")
            assert_bits_equal(img, orc.f_to_byte(out), "packed image")
    r.close()


def test_config1_full_size(rt, orc):
    """The reference's default sphere + plane scene at 512x512, one SS frame (primary, shadow and two mirror bounces
    per pixel): the whole image bit for bit, and the PATH integrator's first sample within its tolerance."""
    sc = rt.scenes.sphere_plane_scene()
    w = h = 512
    r = rt.Renderer(sc, w, h).build()
    img = r.ss_render()
    rays = orc.generate_rays(w, h)
    assert_bits_equal(img, orc.cast_ray_test(sc, rays), "cast_ray_test image")
    assert len(np.unique(img)) > 20
    r.init_path(num_samples=1)
    r.path_render()
    fresh = orc.path_trace(sc, rays, w, h, (0, 0, 0), 1)
    got = r.download(r.path.contents.cl_path_fresh_frame_buffer, np.float32, 4 * w * h).reshape(-1, 4)
    assert (got.view(np.uint32) != fresh.view(np.uint32)).any(axis=1).mean() <= 0.005
    r.close()


def test_debug_views_bit_exact(rt, orc):
    """The reference's two debug probes (spath_raytracer_ray_test / _kd_test, src/spath_raytracer.c:173-235 over
    kdtree_ray_draw / kdtree_test_draw, src/kernels/kdtree.cl:323-384) after a prepass and after five iterations:
    ray origins and hit normals as colours, against the oracle."""
    import numpy as np
    from helpers import assert_bits_equal, oracle_spath_run
    sc = rt.scenes.icosphere_scene(4)
    tree = orc.KdTree(sc.verts4, sc.idx4)
    w, h, seed = 96, 64, 12
    r = rt.Renderer(sc, w, h).build()
    r.init_spath(seed=seed, resolve_interval=0)
    r.spath_prepass(num_iterations=5)
    for iters in (0, 5):
        if iters:
            r.spath_render(iters)
        st = oracle_spath_run(orc, sc, tree, w, h, seed, iters)
        r.L.spath_raytracer_ray_test(r.spath)
        assert_bits_equal(r.output, orc.kd_ray_draw(st.rays), f"ray origins after {iters} iterations")
        r.L.spath_raytracer_kd_test(r.spath)
        want = orc.kd_test_draw(sc, st.results if iters else st.source_results)
        assert_bits_equal(r.output, want, f"hit normals after {iters} iterations")
        assert len(np.unique(want)) > 50
    r.close()
