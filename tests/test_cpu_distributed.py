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


def test_two_rank_band_gather_restores_frame_order(tmp_path):
    import torch.multiprocessing as mp
    from raytracer_amd.strips import band_partition, band_rows
    w, h, world = 5, 48, 2
    out = str(tmp_path / "bands.npy")
    mp.spawn(_band_worker, args=(world, _free_port(), w, h, out), nprocs=world, join=True)
    assert np.array_equal(np.load(out), np.arange(w * h, dtype=np.int32))
    assert band_partition(48, 2, 1) == (8, 8, 16, 24)
    assert band_rows(32, 2, 1)[:9] == [8, 9, 10, 11, 12, 13, 14, 15, 24]
    with pytest.raises(ValueError):
        band_partition(50, 2, 0)
