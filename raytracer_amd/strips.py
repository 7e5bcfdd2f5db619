"""Row-strip partition of a frame over ranks and the framebuffer gather.

Every pixel's path is independent and its RNG streams depend only on the
pixel's coordinates in the full frame, so a rank that renders rows
[y0, y0+rows) of the frame (``raytracer_set_tile``) produces exactly the
pixels the single-GPU render produces there.  The only exchange is one gather
of the resolved strips to rank 0 (RCCL when the tensors live on GPUs, gloo on
CPU); there is no per-iteration communication.
"""
from __future__ import annotations


def strip_rows(height: int, world: int, rank: int) -> tuple[int, int]:
    """(first row, row count) of ``rank``'s strip; rows are spread as evenly
    as possible, earlier ranks take the remainder."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(height, world)
    rows = base + (1 if rank < extra else 0)
    y0 = rank * base + min(rank, extra)
    return y0, rows


def gather_frame(local, width: int, height: int, dist, rank: int, world: int, dst: int = 0):
    """Gather per-rank strips (1-D int32 tensors of width*rows pixels, any
    device the process group supports) into the full frame on ``dst``.
    Returns the [height*width] tensor on ``dst`` and None elsewhere.  Strips
    may differ in size by one row, so the gather is padded to the largest."""
    import torch
    sizes = [strip_rows(height, world, r)[1] * width for r in range(world)]
    pad = max(sizes)
    buf = local
    if local.numel() != pad:
        buf = torch.zeros(pad, dtype=local.dtype, device=local.device)
        buf[:local.numel()] = local
    if world == 1:
        return local.clone()
    out = [torch.empty(pad, dtype=local.dtype, device=local.device) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return torch.cat([out[r][:sizes[r]] for r in range(world)])
