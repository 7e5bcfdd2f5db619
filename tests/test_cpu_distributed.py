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
