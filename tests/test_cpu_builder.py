"""kd-tree builder parity on the CPU: oracle restatement vs the reference's own
builder (oracle/_ref, where built), product builder vs oracle, and both vs the
committed golden blob."""
import hashlib
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _scenes(rt):
    s = rt.scenes
    return {"ico2": s.icosphere_scene(2), "ico4": s.icosphere_scene(4), "soup2k": s.soup_scene(2000),
            "mixed": s.mixed_scene(2), "bundles": s.bundle_scene()}


@pytest.mark.parametrize("name", ["ico2", "ico4", "soup2k", "mixed", "bundles"])
def test_oracle_builder_matches_reference_builder(rt, orc, name):
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref not built (no reference checkout here)")
    sc = _scenes(rt)[name]
    kt = orc.KdTree(sc.verts4, sc.idx4)
    blob, counts, bounds = orc.ref_kd_blob(sc.verts4, sc.idx4)
    assert kt.blob_size == blob.shape[0]
    assert [kt.counts[k] for k in ("nodes_total", "traversal", "leaves", "indices_total", "tris_padded")] == list(counts)
    m = orc.blob_significant_mask(kt.blob, kt.blob_used)
    assert (kt.blob[m] == blob[m]).all()
    assert np.array_equal(bounds[:3], kt.bounds_min) and np.array_equal(bounds[3:], kt.bounds_max)
    assert kt.counts["count_mismatch"] == 0


def test_reference_obj_mesh_if_present(rt, orc):
    """One of the reference's own meshes through all three builders (only where
    the reference checkout and its builder are available)."""
    path = "/root/reference/res/mesh/test.obj"
    if orc.ref_lib() is None or not os.path.exists(path):
        pytest.skip("reference checkout not present")
    v, vn, f = [], [], []
    for line in open(path):
        p = line.split()
        if not p:
            continue
        if p[0] == "v":
            v.append([float(x) for x in p[1:4]])
        elif p[0] == "vn":
            vn.append([float(x) for x in p[1:4]])
        elif p[0] == "f":
            idx = [[int(a) if a else 0 for a in (q.split("/") + ["", ""])[:3]] for q in p[1:]]
            for i in range(1, len(idx) - 1):
                f.append([idx[0], idx[i], idx[i + 1]])
    f = np.array(f)
    sc = rt.scenes._assemble([(np.array(v, np.float32), np.array(vn, np.float32), f[:, :, 0] - 1, f[:, :, 2] - 1, 3)],
                             rt.scenes._materials(rt.scenes._PATH_MATERIALS))
    kt = orc.KdTree(sc.verts4, sc.idx4)
    blob, counts, _ = orc.ref_kd_blob(sc.verts4, sc.idx4)
    m = orc.blob_significant_mask(kt.blob, kt.blob_used)
    assert (kt.blob[m] == blob[m]).all()
    hk = rt.HostKdTree(sc)
    assert (hk.blob == kt.blob[:kt.blob_size]).all()


@pytest.mark.parametrize("case", ["scene:scenes/path_obj.rsc", "mesh:mesh/test3.obj", "mesh:mesh/scene3_1.obj"])
def test_reference_scenes_through_loader_and_all_three_builders(rt, orc, case):
    """The reference's own scene files and its largest mesh (46,966 triangles), read by the library's loader, then
    the REFERENCE's builder against the oracle's and the product's: same serialised tree."""
    root = "/root/reference/res"
    if orc.ref_lib() is None or not os.path.isdir(root):
        pytest.skip("reference checkout not present")
    kind, url = case.split(":")
    if kind == "scene":
        sc = rt.load_scene(url, root)
    else:
        sc = rt.load_scene(None, root, text='{"name":"m","materials":[{"r":1,"g":1,"b":1,"reflectivity":0.5}],'
                                            '"meshes":[{"url":"%s","mat_index":0}],"primitives":{}}' % url)
    assert sc.num_tris > 10000
    kt = orc.KdTree(sc.verts4, sc.idx4)
    blob, counts, bounds = orc.ref_kd_blob(sc.verts4, sc.idx4)
    assert [kt.counts[k] for k in ("nodes_total", "traversal", "leaves", "indices_total", "tris_padded")] == list(counts)
    m = orc.blob_significant_mask(kt.blob, kt.blob_used)
    assert (kt.blob[m] == blob[m]).all()
    hk = rt.HostKdTree(sc)
    assert (hk.blob == kt.blob[:kt.blob_size]).all()


@pytest.mark.parametrize("name", ["ico2", "ico4", "soup2k", "mixed", "bundles"])
@pytest.mark.parametrize("threads", [1, 4])
def test_product_builder_matches_oracle(rt, orc, name, threads):
    sc = _scenes(rt)[name]
    kt = orc.KdTree(sc.verts4, sc.idx4)
    hk = rt.HostKdTree(sc, threads=threads)
    assert hk.blob.shape[0] == kt.blob_size
    assert (hk.blob == kt.blob[:kt.blob_size]).all()
    for k in ("nodes_total", "traversal", "leaves", "indices_total", "tris_padded"):
        assert hk.counts[k] == kt.counts[k]


def _decode_flat(hk):
    """Depth-first walk of the flattened tree: (leaves as index-offset tuples, interior nodes as (axis, split bits))."""
    leaves, interior, used = [], [], set()
    words = hk.leaf_tris.view(np.uint32).reshape(-1)

    def walk(ref):
        if ref & 0x80000000:
            cnt, line = (ref >> 26) & 31, ref & 0x03FFFFFF
            slots = words[16 * line:].reshape(-1)                   # 10 words per slot from the leaf's first line
            first = 0
            if cnt == 31:                                           # header slot: zero-extent triangle + slot count
                assert not slots[:9].any()
                cnt, first = int(slots[9]), 1
            assert cnt == 0 or line + (40 * cnt + 63) // 64 <= hk.leaf_tris.shape[0]
            leaves.append(tuple(int(slots[10 * i + 9]) for i in range(first, cnt)))
            return
        rec, pos = hk.nodes[ref >> 2], ref & 3
        used.add(ref >> 2)
        packed = int(rec[3])
        interior.append(((packed >> (2 * pos)) & 3, int(rec[pos])))
        if pos == 0:       # top of a record: its children are positions 1 / 2 of the same record, or leaves
            walk(int(rec[4]) if packed & 64 else (ref | 1))
            walk(int(rec[6]) if packed & 128 else (ref | 2))
        else:
            assert not packed & (32 << pos)
            walk(int(rec[2 + 2 * pos]))
            walk(int(rec[3 + 2 * pos]))
    import sys
    sys.setrecursionlimit(10000)
    walk(hk.root_ref)
    return leaves, interior, used


def _leaf_refs(hk, used):
    for i in used:
        rec = hk.nodes[i]
        for side in (0, 1):
            words = (4 + 2 * side,) if int(rec[3]) & (64 << side) else (4 + 2 * side, 5 + 2 * side)
            for wd in words:
                if int(rec[wd]) & 0x80000000:
                    yield int(rec[wd])


def _decode_blob(blob):
    """The same from the reference's wire format, which is in depth-first order already."""
    leaves, interior, off = [], [], 0
    while off < blob.shape[0]:
        t = int(blob[off])
        if t == 2:
            interior.append((int(blob[off + 1]), int(blob[off + 4:off + 8].view(np.uint32)[0])))
            off += 32
        elif t == 1:
            n = int(blob[off + 4:off + 8].view(np.uint32)[0])
            leaves.append(tuple(int(x) for x in blob[off + 16:off + 16 + 4 * n].view(np.uint32)))
            off += 16 + 4 * n + (4 if n % 2 else 0)
        else:
            break
    return leaves, interior


def test_leaves_longer_than_the_descriptor_count(rt, orc):
    """Stacks of 45 and 33 coincident triangles: one leaf each, with a header slot (a triangle of zero extent whose
    last word is the slot count) in front of the triangles."""
    sc = _scenes(rt)["bundles"]
    hk = rt.HostKdTree(sc)
    kt = orc.KdTree(sc.verts4, sc.idx4)
    flat_leaves, flat_interior, used = _decode_flat(hk)
    blob_leaves, blob_interior = _decode_blob(kt.blob[:kt.blob_used])
    assert flat_leaves == blob_leaves and flat_interior == blob_interior
    assert sorted(len(l) for l in flat_leaves if len(l) >= 31) == [33, 45]
    long_refs = [r for r in _leaf_refs(hk, used) if (r >> 26) & 31 == 31]
    assert len(long_refs) == 2
    words = hk.leaf_tris.view(np.uint32).reshape(-1)
    for r in long_refs:
        line = r & 0x03FFFFFF
        assert not words[16 * line:16 * line + 9].any() and int(words[16 * line + 9]) in (46, 34)


@pytest.mark.parametrize("name", ["ico4", "soup2k"])
def test_flattened_tree_has_the_same_leaves_in_the_same_order(rt, orc, name):
    sc = _scenes(rt)[name]
    hk = rt.HostKdTree(sc)
    kt = orc.KdTree(sc.verts4, sc.idx4)
    flat_leaves, flat_interior, used = _decode_flat(hk)
    blob_leaves, blob_interior = _decode_blob(kt.blob[:kt.blob_used])
    assert flat_leaves == blob_leaves
    assert flat_interior == blob_interior and len(flat_interior) == hk.counts["traversal"]
    assert len(used) < hk.counts["traversal"]                  # two levels per record
    # lines of four records holding connected pieces of the tree: most records are referred to from a record of
    # their own line; slots nothing refers to hold zeros
    assert hk.nodes.shape[0] % 4 == 0 and len(used) > 0.8 * hk.nodes.shape[0]
    free = np.setdiff1d(np.arange(hk.nodes.shape[0]), sorted(used))
    assert not hk.nodes[free].any()
    parents = {}
    for i in used:
        rec = hk.nodes[i]
        for side in (0, 1):
            if not int(rec[3]) & (64 << side):
                for wd in (4 + 2 * side, 5 + 2 * side):
                    if not int(rec[wd]) & 0x80000000:
                        parents[int(rec[wd]) >> 2] = i
    together = sum(parents[i] // 4 == i // 4 for i in used if i in parents)
    assert together > 0.4 * len(used)               # three of every four at best
    # a leaf of two lines (2 or 3 triangles) lies in one aligned pair of lines
    for ref in _leaf_refs(hk, used):
        cnt, line = (ref >> 26) & 31, ref & 0x03FFFFFF
        if cnt in (2, 3):
            assert line % 2 == 0
    # precomputed edges are exactly v1-v0, v2-v0
    first = hk.leaf_tris.reshape(-1)[:10]
    off = int(first[9:10].view(np.uint32)[0])
    a, b, c = (sc.verts4[sc.idx4[off + i, 0], :3] for i in range(3))
    assert np.array_equal(first[0:3], a)
    assert np.array_equal(first[3:6], b - a)
    assert np.array_equal(first[6:9], c - a)


def test_single_leaf_tree_and_empty_scene(rt, orc):
    sc = rt.scenes._assemble([rt.scenes._quad((0, 0, -3), (1, 0, -3), (1, 1, -3), (0, 1, -3), (0, 0, 1)) + (0,)],
                             rt.scenes._materials(rt.scenes._PATH_MATERIALS))
    hk = rt.HostKdTree(sc)
    assert hk.root_ref & 0x80000000 and hk.counts["leaves"] == 1 and hk.counts["traversal"] == 0
    kt = orc.KdTree(sc.verts4, sc.idx4)
    assert (hk.blob == kt.blob[:kt.blob_size]).all()


def test_golden_blob(rt, orc):
    """Golden vector: the serialised tree of the ico2 scene as produced by the
    REFERENCE's builder (tests/golden/make_golden.py wrote it from oracle/_ref)."""
    g = np.load(os.path.join(GOLDEN, "kd_ico2.npz"))
    sc = rt.scenes.icosphere_scene(2)
    assert hashlib.sha256(sc.verts4.tobytes() + sc.idx4.tobytes()).hexdigest() == str(g["scene_sha256"])
    kt = orc.KdTree(sc.verts4, sc.idx4)
    m = g["mask"].astype(bool)
    assert (kt.blob[:kt.blob_size][m] == g["ref_blob"][m]).all()
    hk = rt.HostKdTree(sc)
    assert (hk.blob[m] == g["ref_blob"][m]).all()
