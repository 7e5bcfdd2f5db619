"""The C-ABI library on a machine without a GPU: it loads, exports every
symbol the headers declare, and its struct layouts match the binding.  No
compute entry point is called here."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_abi_matches(rt):
    L = rt.lib()                      # raises on any size/offset mismatch
    assert L.rt_abi_query(b"sizeof:mesh") == 128
    assert L.rt_abi_query(b"sizeof:material") == 32
    assert L.rt_abi_query(b"sizeof:sphere") == 32
    assert L.rt_abi_query(b"sizeof:plane") == 48
    assert L.rt_abi_query(b"sizeof:ray") == 32
    assert L.rt_abi_query(b"sizeof:kd_tree_collision_result") == 16
    assert L.rt_abi_query(b"sizeof:_skd_tree_traversal_node") == 32
    assert L.rt_abi_query(b"sizeof:_skd_tree_leaf_node") == 16
    assert L.rt_abi_query(b"offsetof:mesh.material_index") == 104
    assert rt.scenes.MESH_DT.itemsize == 128 and rt.scenes.PLANE_DT.itemsize == 48


def test_every_declared_symbol_is_exported(rt):
    L = rt.lib()
    declared = set()
    for name in os.listdir(os.path.join(ROOT, "include")):
        text = open(os.path.join(ROOT, "include", name)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"^[A-Za-z_][\w \*]*?\b(\w+)\s*\([^;{]*\)\s*;", text, flags=re.M):
            declared.add(m.group(1))
    declared -= {"defined", "__attribute__", "aligned"}
    assert len(declared) > 60
    missing = sorted(s for s in declared if not hasattr(L, s))
    assert not missing, missing
    assert not [s for s in rt.EXPORTED_SYMBOLS if not hasattr(L, s)]


def test_host_geometry_helpers(rt):
    import ctypes as C
    L = rt.lib()
    box = rt.AABB()
    box.max[:] = (2, 3, 4, 0)
    box.min[:] = (0, 1, 1, 0)
    L.AABB_surface_area.restype = C.c_float
    L.AABB_surface_area.argtypes = [rt.AABB]
    assert L.AABB_surface_area(box) == 2 * (2 * 2 + 2 * 3 + 2 * 3)
    left, right = rt.AABB(), rt.AABB()
    L.AABB_divide_world.argtypes = [rt.AABB, C.c_uint8, C.c_float, C.POINTER(rt.AABB), C.POINTER(rt.AABB)]
    L.AABB_divide_world(box, 1, 2.0, C.byref(left), C.byref(right))
    assert left.max[1] == 2.0 and right.min[1] == 2.0 and left.min[1] == 1.0 and right.max[1] == 3.0
    L.AABB_ilerp.restype = C.c_float
    L.AABB_ilerp.argtypes = [rt.AABB, C.c_uint8, C.c_float]
    assert L.AABB_ilerp(box, 0, 0.5) == 0.25


def test_strip_partition():
    from raytracer_amd.strips import strip_rows
    for h, w in ((2048, 8), (2160, 7), (5, 3), (3, 4)):
        rows = [strip_rows(h, w, r) for r in range(w)]
        assert rows[0][0] == 0 and sum(n for _, n in rows) == h
        for (a, n), (b, _) in zip(rows, rows[1:]):
            assert a + n == b
        assert max(n for _, n in rows) - min(n for _, n in rows) <= 1


def test_no_product_import_of_the_oracle():
    """The product path must never route through the oracle."""
    pkg = os.path.join(ROOT, "raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hpp", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "oracle_rt" not in text, f


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` run plainly starts its own ranks -- after checking, without touching a GPU, that N
    are visible.  With fewer (none in a CPU container) it exits non-zero with a message and prints no result line."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "RT_BENCH_SHARE_GPU")}
    import torch
    if torch.cuda.device_count() >= 64:
        return
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64", "--workload", "tiny"], cwd=root,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and not p.stdout.strip()
    assert b"GPU(s) are visible" in p.stderr


def test_plane_distance_by_reciprocal_is_the_correctly_rounded_quotient(tmp_path):
    """The kernels compute (b - o) / d for most rays as (b - o) * RN(1/d) corrected by two fused multiply-adds
    (rt_kernels.hip: walk_descend).  tests/tools/check_plane_distance_division.c checks that identity against the
    division on random operands; here 2e8 of them (the committed figure is 6e10 with none wrong)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "check")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-mfma",
                           os.path.join(root, "tests", "tools", "check_plane_distance_division.c"), "-o", exe, "-lm"])
    out = subprocess.run([exe, "200000000"], stdout=subprocess.PIPE, timeout=300).stdout.decode()
    assert "bad=0" in out, out
