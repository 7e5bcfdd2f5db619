"""Row-strip partition of a frame over ranks and the framebuffer gather.

Every pixel's path is independent and its RNG streams depend only on the
pixel's coordinates in the full frame, so a rank that renders rows
[y0, y0+rows) of the frame (``raytracer_set_tile``) produces exactly the
pixels the single-GPU render produces there.  The only exchange is one gather
of the resolved strips to rank 0 (RCCL when the tensors live on GPUs, gloo on
CPU); there is no per-iteration communication.
"""
from __future__ import annotations

BAND_ROWS = 8      # rows per band of the interleaved partition (one screen tile of the pixel sort)


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


def band_partition(height: int, world: int, rank: int, band: int = BAND_ROWS) -> tuple[int, int, int, int]:
    """Interleaved partition for load balance (the centre of a frame is usually
    the expensive part): the frame is cut into bands of ``band`` rows dealt to
    the ranks in turn.  Returns (first row, band, stride, local row count) for
    ``raytracer_set_tile_bands``.  ``height`` must be a multiple of
    ``world * band``."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    if height % (world * band):
        raise ValueError(f"height {height} is not a multiple of {world} ranks x {band} rows")
    return rank * band, band, world * band, height // world


def band_rows(height: int, world: int, rank: int, band: int = BAND_ROWS) -> list[int]:
    """Frame rows of ``rank``'s local rows 0, 1, ... under band_partition."""
    first, band, stride, rows = band_partition(height, world, rank, band)
    return [first + (j // band) * stride + j % band for j in range(rows)]


def gather_bands(local, width: int, height: int, dist, rank: int, world: int, dst: int = 0, band: int = BAND_ROWS):
    """Gather the ranks' interleaved rows (1-D int32 tensors of width*rows
    pixels) into the full frame on ``dst``; None elsewhere."""
    import torch
    if world == 1:
        return local.clone()
    out = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, out, dst=dst)
    if rank != dst:
        return None
    # rank r holds bands r, r + world, ...: stacking [world, bands per rank, band*width] and swapping the first two
    # axes puts the bands in frame order
    per_rank = height // (world * band)
    stacked = torch.stack(out).view(world, per_rank, band * width)
    return stacked.transpose(0, 1).reshape(-1)
