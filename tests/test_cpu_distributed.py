"""The N>1 path on CPU: two gloo ranks each render their row strip (with the
oracle standing in for the GPU renderer), the resolved strips are gathered to
rank 0 and must equal the single-process frame."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, iters, seed, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import oracle as orc
    from helpers import oracle_spath_run
    from raytracer_amd import scenes
    from raytracer_amd.strips import gather_frame, strip_rows
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    sc = scenes.icosphere_scene(3)
    tree = orc.KdTree(sc.verts4, sc.idx4)
    y0, rows = strip_rows(h, world, rank)
    st = oracle_spath_run(orc, sc, tree, w, rows, seed, iters, y0=y0, full_h=h)
    local = torch.from_numpy(st.resolve().view(np.int32).copy())
    frame = gather_frame(local, w, h, dist, rank, world)
    if rank == 0:
        np.save(out_path, frame.numpy().view(np.uint32))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("h", [24, 25])
def test_two_rank_strips_equal_single_frame(tmp_path, orc, rt, h):
    import torch.multiprocessing as mp
    from helpers import oracle_spath_run
    w, iters, seed, world = 32, 5, 9, 2
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, iters, seed, out), nprocs=world, join=True)
    got = np.load(out)
    sc = rt.scenes.icosphere_scene(3)
    tree = orc.KdTree(sc.verts4, sc.idx4)
    want = oracle_spath_run(orc, sc, tree, w, h, seed, iters).resolve()
    assert np.array_equal(got, want)


def _band_worker(rank, world, port, w, h, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from raytracer_amd.strips import band_rows, gather_bands
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    rows = band_rows(h, world, rank)
    # every pixel carries its own frame coordinates
    local = torch.tensor([[y * w + x for x in range(w)] for y in rows], dtype=torch.int32).reshape(-1)
    frame = gather_bands(local, w, h, dist, rank, world)
    if rank == 0:
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("w,h,world", [(5, 48, 2), (3, 56, 3), (2, 61, 3), (1, 2160, 4), (1, 2160, 8)])
def test_band_gather_restores_frame_order(tmp_path, w, h, world):
    """Even and ragged shares (bands not a multiple of the ranks, a short last band), up to the 2160 rows of the
    3840x2160 configuration on 4 and 8 ranks: the gathered frame is in row order."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "bands.npy")
    mp.spawn(_band_worker, args=(world, _free_port(), w, h, out), nprocs=world, join=True)
    assert np.array_equal(np.load(out), np.arange(w * h, dtype=np.int32))


def test_band_partition_shares():
    from raytracer_amd.strips import band_partition, band_rows
    assert band_partition(48, 2, 1) == (8, 8, 16, 24)
    assert band_rows(32, 2, 1)[:9] == [8, 9, 10, 11, 12, 13, 14, 15, 24]
    # 2160 rows = 270 bands: six ranks of eight hold 34 bands, two hold 33
    assert [band_partition(2160, 8, r)[3] for r in range(8)] == [272] * 6 + [264] * 2
    assert [band_partition(2160, 4, r)[3] for r in range(4)] == [544, 544, 536, 536]
    # a short last band belongs to the rank whose turn it is
    assert [band_partition(50, 2, r)[3] for r in range(2)] == [26, 24]
    assert band_rows(50, 2, 0)[-3:] == [39, 48, 49]
    for h, world in [(2160, 8), (2160, 4), (56, 3), (61, 3), (9, 2), (2048, 8)]:
        rows = sorted(y for r in range(world) for y in band_rows(h, world, r))
        assert rows == list(range(h))
    with pytest.raises(ValueError):
        band_partition(8, 2, 0)        # one band, two ranks
