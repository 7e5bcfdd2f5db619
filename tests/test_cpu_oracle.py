"""Oracle self-checks that need no GPU: golden vectors, properties the domain
offers, numeric helpers."""
import hashlib
import os

import numpy as np
import pytest

from helpers import assert_bits_equal, hits_equal, oracle_spath_run

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_sincos_within_two_ulp(orc):
    theta = np.concatenate([np.linspace(0, 2 * np.pi, 200001), np.random.RandomState(0).uniform(0, 2 * np.pi, 200000)]).astype(np.float32)
    s, c = orc.sincos(theta)
    rs, rc = np.sin(theta.astype(np.float64)), np.cos(theta.astype(np.float64))
    ulp_s = np.abs(s - rs) / np.spacing(np.maximum(np.abs(rs), 2.0 ** -20).astype(np.float32))
    ulp_c = np.abs(c - rc) / np.spacing(np.maximum(np.abs(rc), 2.0 ** -20).astype(np.float32))
    assert ulp_s.max() <= 2.0 and ulp_c.max() <= 2.0


def test_primary_rays(orc):
    w, h = 64, 32
    r = orc.generate_rays(w, h)
    assert np.all(r[:, :4] == 0)
    d = r[:, 4:7].astype(np.float64)
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-6)
    assert (d[:, 2] < 0).all()
    # strip == rows of the full frame
    s = orc.generate_rays(w, 8, y0=16, full_h=h)
    assert_bits_equal(s, r.reshape(h, w, 8)[16:24].reshape(-1, 8), "strip rays")
    # horizontal symmetry of the pinhole camera
    dd = r.reshape(h, w, 8)[:, :, 4]
    assert np.allclose(dd, -dd[:, ::-1], atol=1e-6)


@pytest.mark.parametrize("name,n", [("soup", 2000), ("ico", 4)])
def test_traversal_vs_brute_force(rt, orc, name, n):
    """Same hit/miss set as an exact nearest-hit search; where the triangle
    differs the kd result is the FARTHER one (the reference's leaf early exit
    takes the first accepted triangle, src/kernels/kdtree.cl:284-296)."""
    sc = rt.scenes.soup_scene(n) if name == "soup" else rt.scenes.icosphere_scene(n)
    tree = orc.KdTree(sc.verts4, sc.idx4)
    rays = orc.generate_rays(96, 96)
    kd, st = tree.intersect(rays, stats=True)
    bf = orc.brute_nearest(sc, rays)
    # the traversal clips rays to the %f-rounded scene box; brute force does not
    assert np.array_equal(kd["t"] != 0, bf["t"] != 0)
    same = kd["tri"] == bf["tri"]
    hit = kd["t"] != 0
    assert_bits_equal(kd["t"][same & hit], bf["t"][same & hit], "t of agreeing hits")
    diff = hit & ~same
    assert (kd["t"][diff] >= bf["t"][diff]).all()
    assert diff.sum() <= 0.05 * max(hit.sum(), 1)
    assert st["rays"] == rays.shape[0] and st["max_stack"] <= 64


def test_golden_spath(rt, orc):
    g = np.load(os.path.join(GOLDEN, "spath_ico3.npz"))
    sc = rt.scenes.icosphere_scene(3)
    assert hashlib.sha256(sc.verts4.tobytes() + sc.idx4.tobytes()).hexdigest() == str(g["scene_sha256"])
    tree = orc.KdTree(sc.verts4, sc.idx4)
    st = oracle_spath_run(orc, sc, tree, 48, 32, 11, 6)
    hits_equal(st.source_results, g["source_results"], "primary hits")
    assert_bits_equal(st.out_tex, g["out_tex"], "out_tex")
    assert_bits_equal(st.progress, g["progress"], "progress")
    assert_bits_equal(st.rays, g["rays"], "rays")
    assert_bits_equal(st.resolve(), g["image"], "image")


def test_spath_strip_equals_full_frame(rt, orc):
    sc = rt.scenes.icosphere_scene(3)
    tree = orc.KdTree(sc.verts4, sc.idx4)
    w, h = 40, 24
    full = oracle_spath_run(orc, sc, tree, w, h, 5, 5)
    strip = oracle_spath_run(orc, sc, tree, w, 8, 5, 5, y0=8, full_h=h)
    assert_bits_equal(strip.out_tex, full.out_tex.reshape(h, w, 4)[8:16].reshape(-1, 4), "strip out_tex")


def test_first_iteration_is_a_sky_sample(rt, orc):
    """The reference wipes the ray buffer at the end of the prepass
    (src/spath_raytracer.c:432-433): the first iteration traces null rays, every
    pixel finishes one sample of mask*sky and restarts from its primary hit."""
    sc = rt.scenes.icosphere_scene(3)
    tree = orc.KdTree(sc.verts4, sc.idx4)
    st = oracle_spath_run(orc, sc, tree, 32, 32, 3, 1)
    assert (st.results["t"] == 0).all()
    assert (st.out_tex[:, 3] == 1).all()
    assert (st.progress[:, 0] == 1).all() and (st.progress[:, 1] == 0).all()


def test_libm_trig_mode_stays_close(rt, orc):
    """What fixing one sin/cos implementation is worth: the same render with the
    C library's sinf/cosf differs by a small RMSE only (tolerance 2e-3 per
    channel on the accumulated mean, as the measurement plan states)."""
    import subprocess, sys, json
    code = ("import sys,json,numpy as np; sys.path.insert(0,%r); sys.path.insert(0,%r+'/tests');"
            "import oracle as orc; from helpers import oracle_spath_run; from raytracer_amd import scenes;"
            "sc=scenes.icosphere_scene(3); t=orc.KdTree(sc.verts4,sc.idx4); st=oracle_spath_run(orc,sc,t,48,32,11,24);"
            "m=st.out_tex[:,:3]/np.maximum(st.out_tex[:,3:4],1); print(json.dumps(m.tolist()))")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for mode in ("0", "1"):
        env = dict(os.environ, ORC_LIBM_TRIG=mode)
        outs.append(np.array(json.loads(subprocess.check_output([sys.executable, "-c", code % (root, root)], env=env))))
    rmse = float(np.sqrt(np.mean((outs[0] - outs[1]) ** 2)))
    assert rmse < 2e-3, rmse


def test_xorshift_and_resolve(orc):
    d = np.array([0, 1, 0x80000000, 0xFFFFFFFF, 12345], np.uint32)
    e = d.copy()
    orc.xorshift(e)
    want = (((d.astype(np.uint64) << 1) | (d.astype(np.uint64) >> 31)) + 1) & 0xFFFFFFFF
    assert np.array_equal(e, want.astype(np.uint32))
    tex = np.array([[2, 1, 0.5, 2], [0, 0, 0, 0], [10, -1, 3, 4]], np.float32)
    img = orc.lib()
    st = np.zeros(3, np.uint32)
    import ctypes as C
    img.orc_resolve_avg(st.ctypes.data_as(C.c_void_p), tex.ctypes.data_as(C.c_void_p), C.c_uint32(3))
    assert st[0] == (255 << 24) | (255 << 16) | (127 << 8) | 63
    assert st[1] == 0
    assert st[2] == (255 << 24) | (255 << 16) | (0 << 8) | 191


def test_golden_brute_force_integrators(rt, orc):
    """The oracle's SS image and first PATH sample of the sphere + plane scene against their frozen copies."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ss_sphere_plane.npz"))
    sc = rt.scenes.sphere_plane_scene()
    rays = orc.generate_rays(96, 64)
    assert np.array_equal(orc.cast_ray_test(sc, rays), g["image"])
    got = orc.path_trace(sc, rays, 96, 64, (0, 0, 0), 1)
    # seeds go through the C library's sin() of huge arguments: allow the odd pixel to differ between libm builds
    assert (got.view(np.uint32) != g["first_sample"].view(np.uint32)).any(axis=1).mean() <= 0.005


def test_primary_visibility_matches_the_reference_stored_render(rt, orc):
    """The one output of the reference itself that this path can be held against: its stored 1920x1080 frame of
    res/scenes/path_obj2.rsc (renders/3_custom_2046.PNG; fixture and what is taken from it:
    tests/golden/make_ref_render_fixture.py).  Ray generation, scene assembly and first-hit traversal of the
    oracle reproduce the frame's emitter silhouette (276,947 pixels) pixel for pixel and cover every pixel that
    shows the objects' colour."""
    from helpers import load_ref_render_fixture, check_against_stored_render, oracle_spath_run
    sc, w, h, emitter, objects = load_ref_render_fixture(rt)
    assert emitter.sum() == 276947
    tree = orc.KdTree(sc.verts4, sc.idx4)
    st = oracle_spath_run(orc, sc, tree, w, h, 1, 0)
    check_against_stored_render(sc, st.source_results, emitter, objects)


def test_builtin_modes_differ_only_in_rounding(rt, orc):
    """The oracle's two definitions of dot / cross / min / max -- ROCm's OpenCL library's (the default: what the
    reference's compiled kernels compute on the GPU, bit for bit, and what the library is built to) and the plain
    one -- are the same algorithm: on camera and random rays they find the same triangles, t agrees to rounding,
    and some low bits do differ."""
    sc = rt.scenes.icosphere_scene(3)
    tree = orc.KdTree(sc.verts4, sc.idx4)
    rays = orc.generate_rays(96, 96)
    rng = np.random.RandomState(3)
    extra = np.zeros((4096, 8), np.float32)
    extra[:, 0:3] = rng.uniform(-20, 20, size=(4096, 3))
    d = rng.normal(size=(4096, 3)).astype(np.float32)
    extra[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.ascontiguousarray(np.concatenate([rays, extra]))
    rocm = tree.intersect(rays)
    orc.set_builtins("plain")
    try:
        plain = tree.intersect(rays)
    finally:
        orc.set_builtins("rocm")
    again = tree.intersect(rays)
    assert np.array_equal(rocm.view(np.uint32), again.view(np.uint32))          # the switch leaves nothing behind
    assert (plain["tri"] != rocm["tri"]).mean() <= 1e-4
    same = plain["tri"] == rocm["tri"]
    assert np.abs(plain["t"][same] - rocm["t"][same]).max() <= 1e-5 * np.abs(plain["t"]).max()
    assert (plain["t"].view(np.uint32) != rocm["t"].view(np.uint32)).any()
    assert tree.ray_stack(rays).max() <= 16


def test_reference_kernel_build_matches_the_reference_host_calls():
    """Where oracle/_ref holds the code objects compiled from the reference's kernel files, their metadata must
    show the argument lists the reference's host code sets (counts of clSetKernelArg calls in
    src/spath_raytracer.c:143-155,258-263,287-303,325-345, src/raytracer.c:187-190, src/ss_raytracer.c:14-25)."""
    import json
    import os
    base = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
    path = os.path.join(base, "refcl_ico4_strict.json")
    if not os.path.exists(path):
        pytest.skip("reference kernels not built here")
    meta = json.load(open(path))
    want = {"kdtree_intersection": 8, "segmented_path_trace_init": 13, "segmented_path_trace": 14, "generate_rays": 4,
            "xorshift_batch": 1, "f_buffer_to_byte_buffer_avg": 5, "cast_ray_test": 12}
    for name, count in want.items():
        args = [a for a in meta["kernels"][name]["args"] if not a["kind"].startswith("hidden")]
        assert len(args) == count, name
    images = [a for a in meta["kernels"]["kdtree_intersection"]["args"] if a["kind"] == "image"]
    assert len(images) == 2 and all(a["size"] == 8 for a in images)
    assert meta["macros"]["BLOCKSIZE_X"] == "64" and meta["macros"]["SCENE_MIN"].count(",") == 2
    for flavour in ("default", "strict"):
        assert os.path.getsize(os.path.join(base, f"refcl_ico4_{flavour}.hsaco")) > 50_000


def test_hsa_launcher_exports(orc):
    """oracle/libhsa_runner.so (the launcher for those code objects) loads and exports its entry points; nothing
    is called here, a GPU is needed for that."""
    import ctypes as C
    import os
    path = os.path.join(os.path.dirname(orc.__file__), "libhsa_runner.so")
    if not os.path.exists(path):
        pytest.skip("no HSA headers here: launcher not built")
    L = C.CDLL(path)
    for name in ("hr_init", "hr_alloc", "hr_free", "hr_h2d", "hr_d2h", "hr_fill32", "hr_load", "hr_kernel_info",
                 "hr_dispatch", "hr_last_error", "hr_agent_name"):
        assert hasattr(L, name), name


def test_oracle_against_outputs_of_the_reference_kernels(rt, orc):
    """tests/golden/refkernel_ico4.npz holds inputs and outputs of the REFERENCE's own kernels -- its OpenCL C files
    compiled unmodified for gfx950 and run on an MI355X (make_refkernel_fixture.py beside it).  The oracle must
    reproduce them: kdtree_intersection bit for bit on all 22,464 rays (primary, wiped, bounce, hand-made);
    xorshift_batch bit for bit; segmented_path_trace_init and two iterations of segmented_path_trace from the
    frozen pre-step buffers: counters and RNG words equal, floats to 1e-5 of their scale (the shading goes through
    normalize / sin / cos, which ROCm's OpenCL library and the oracle define differently)."""
    import ctypes as C
    z = np.load(os.path.join(GOLDEN, "refkernel_ico4.npz"))
    sc = rt.scenes.icosphere_scene(4)
    assert hashlib.sha256(sc.verts4.tobytes() + sc.idx4.tobytes()).hexdigest() == str(z["scene_sha256"])
    tree = orc.KdTree(sc.verts4, sc.idx4)
    w, h = int(z["width"]), int(z["height"])
    n = w * h
    rv = z["random_values"]

    hits_equal(tree.intersect(z["rays"]), z["hits"], "oracle vs reference kernel")
    assert (z["hits"]["t"] != 0).sum() > 8000 and tree.ray_stack(z["rays"]).max() <= 16

    x = z["xorshift_in"].copy()
    orc.xorshift(x)
    assert_bits_equal(x, z["xorshift_out"], "xorshift_batch")

    def close(got, want, what):
        got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
        scale = max(1.0, float(np.abs(want[np.isfinite(want)]).max()))
        assert np.abs(got - want).max() <= 1e-5 * scale, f"{what}: {np.abs(got - want).max()} (scale {scale})"

    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L = orc.lib()
    st = orc.SpathState(sc, tree, w, h)
    st.rays = np.ascontiguousarray(z["rays"][:n]).copy()
    st.results = np.ascontiguousarray(z["init_results"]).copy()
    L.orc_spath_init(p(st.out_tex), p(st.rays), p(st.origin_rays), p(st.results), p(st.source_results), p(st.progress),
                     p(sc.idx4), p(sc.normals4), p(sc.meshes), p(sc.materials), C.c_uint32(w), C.c_uint32(h),
                     C.c_uint32(0), C.c_uint32(int(rv[0])))
    assert_bits_equal(st.origin_rays, z["init_origin_rays"], "init: origin rays")
    hits_equal(st.source_results, z["source_results"], "init: source results")
    assert_bits_equal(st.progress[:, 0:2], z["init_progress"][:, 0:2], "init: counters")
    close(st.progress[:, 4:7].view(np.float32), z["init_progress"][:, 4:7].view(np.float32), "init: mask")
    close(st.rays, z["init_rays"], "init: first-bounce rays")
    close(st.out_tex, z["init_out_tex"], "init: out_tex")

    finished = 0
    for k in (2, 5):
        st = orc.SpathState(sc, tree, w, h)
        st.out_tex = z[f"step_pre_out_tex_{k}"].copy(); st.rays = z[f"step_pre_rays_{k}"].copy()
        st.progress = z[f"step_pre_progress_{k}"].copy(); st.random = z[f"step_pre_random_{k}"].copy()
        st.origin_rays = z["origin_rays"].copy(); st.source_results = z["source_results"].copy()
        st.results = z[f"step_pre_results_{k}"].copy()
        orc.xorshift(st.random)
        L.orc_spath_trace(p(st.out_tex), p(st.rays), p(st.origin_rays), p(st.results), p(st.source_results),
                          p(st.progress), p(st.random), p(sc.idx4), p(sc.normals4), p(sc.meshes), p(sc.materials),
                          C.c_uint32(n), C.c_uint32(int(rv[k])), C.c_int(4))
        post = z[f"step_post_progress_{k}"]
        assert_bits_equal(st.random, z[f"step_post_random_{k}"], f"iteration {k}: random")
        assert_bits_equal(st.progress[:, 0:2], post[:, 0:2], f"iteration {k}: counters")
        close(st.progress[:, 4:7].view(np.float32), post[:, 4:7].view(np.float32), f"iteration {k}: mask")
        close(st.progress[:, 8:11].view(np.float32), post[:, 8:11].view(np.float32), f"iteration {k}: accumulated colour")
        close(st.rays, z[f"step_post_rays_{k}"], f"iteration {k}: rays")
        close(st.out_tex, z[f"step_post_out_tex_{k}"], f"iteration {k}: out_tex")
        finished = max(finished, int(z[f"step_post_out_tex_{k}"][:, 3].max()))
    assert finished >= 2


def test_brute_force_integrators_against_outputs_of_the_reference_kernels(rt, orc):
    """tests/golden/refkernel_brute.npz: cast_ray_test images and a first path_trace launch from the REFERENCE's own
    kernels on an MI355X (make_refkernel_fixture.py).  Those kernels go through the built-ins normalize, sqrt, pow
    and sin of ROCm's OpenCL library, so silhouette / shadow-edge pixels may fall the other way: stated tolerance
    99.5 % of the packed pixels equal for cast_ray_test, 99 % of the pixels within 1e-4 for path_trace."""
    z = np.load(os.path.join(GOLDEN, "refkernel_brute.npz"))
    for tag, sc in (("spheres", rt.scenes.sphere_plane_scene()), ("mixed2", rt.scenes.mixed_scene(2))):
        w, h = (int(v) for v in z[f"{tag}_size"])
        rays = orc.generate_rays(w, h)
        got = orc.cast_ray_test(sc, rays)
        want = z[f"{tag}_cast_ray_test"]
        assert (got == want).mean() >= 0.995, f"{tag}: {(got == want).mean():.4f} of pixels equal"
        assert len(np.unique(want)) > 20
        if tag == "mixed2":
            fresh = orc.path_trace(sc, rays, w, h, (0, 0, 0), 1)
            want = z["mixed2_path_trace_first"]
            assert (np.abs(fresh - want).max(axis=1) < 1e-4).mean() >= 0.99
            assert float(np.sqrt(np.mean((fresh - want) ** 2))) < 1e-2


def test_reference_kernel_prelude_is_what_the_reference_host_would_print():
    """oracle/build_ref_kernels.py hands the macro prelude to the compiler as -D options.  Here the prelude is
    rebuilt from the TEXT of the reference's host code (the MACRO_GEN lines, the two sprintf formats, the os_macro
    of the non-Windows branch and the order of the macros[] array, src/raytracer.c:38-70; MACRO_GEN itself,
    include/parallel.h:7-10, stringizes its format, which collapses runs of blanks) with the scene's values, and
    must be character for character the `#define NAME VALUE` lines the -D options stand for.  Needs the reference
    checkout (read as text); skipped where it is not mounted."""
    import os
    import re
    import sys
    src = "/root/reference/src/raytracer.c"
    if not os.path.exists(src):
        pytest.skip("reference checkout not mounted")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    import build_ref_kernels as brk
    import raytracer_amd.scenes as S

    text = open(src).read()
    body = text[text.index("void raytracer_build_kernels"):text.index("load_program_raw(rctx->rcl")]
    gen = open("/root/reference/include/parallel.h").read()
    assert re.search(r'sprintf\(n, "#define " #t, v\)', gen)               # the stringized format follows "#define "
    formats = {}                                                             # variable -> (format, [value expressions])
    for var, fmt, val in re.findall(r"MACRO_GEN\((\w+),\s*(.+?),\s*([^,]+?),\s*num_macros\);", body):
        formats[var] = ("#define " + " ".join(fmt.split()), [val.strip()])
    for var, fmt, args in re.findall(r'sprintf\((\w+), "([^"]+)",\s*((?:[^;]+?))\);', body):
        formats[var] = (fmt, [a.strip() for a in args.split(",")])
    other = re.search(r'#else\s*char os_macro\[\] = "([^"]+)";', body)
    formats["os_macro"] = (other.group(1), [])
    order = [v.strip() for v in re.search(r"char\* macros\[\]\s*=\s*\{([^}]+)\}", body).group(1).split(",")]
    assert len(order) == 10 and set(order) == set(formats)

    for sc in (S.icosphere_scene(4), S.mixed_scene(2), S.sphere_plane_scene()):
        used = sc.verts4[np.unique(sc.idx4[:, 0]), :3] if sc.idx4.shape[0] else np.zeros((1, 3), np.float32)
        lo, hi = used.min(axis=0), used.max(axis=0)
        values = {"rctx->stat_scene->num_spheres": int(sc.spheres.shape[0]),
                  "rctx->stat_scene->num_planes": int(sc.planes.shape[0]),
                  "rctx->stat_scene->num_mesh_indices": int(sc.idx4.shape[0]),
                  "rctx->stat_scene->num_meshes": int(sc.meshes.shape[0]),
                  "rctx->stat_scene->num_materials": int(sc.materials.shape[0]),
                  "rctx->rcl->simt_size": brk.SIMT_WIDTH, "rctx->rcl->num_simt_per_multiprocessor": 1}
        for k in range(3):
            values[f"rctx->stat_scene->kdt->bounds.min[{k}]"] = float(lo[k])    # float -> double, as C's varargs
            values[f"rctx->stat_scene->kdt->bounds.max[{k}]"] = float(hi[k])
        reference_prelude = "\n".join(formats[v][0] % tuple(values[a] for a in formats[v][1]) for v in order)
        macros = brk.scene_macros(sc)
        ours = {f"#define {k} {v}" for k, v in macros.items()}
        assert ours == set(reference_prelude.split("\n")), (sorted(ours), reference_prelude)
        # and in the reference's own order, as one text
        by_name = {line.split()[1]: line for line in ours}
        assert "\n".join(by_name[line.split()[1]] for line in reference_prelude.split("\n")) == reference_prelude
