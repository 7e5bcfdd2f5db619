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
    the ranks in turn -- rank r renders bands r, r + world, r + 2 world, ...
    Returns (first row, band, stride, local row count) for
    ``raytracer_set_tile_bands``.  The shares may be ragged: when the number of
    bands is not a multiple of ``world`` the first ranks hold one band more than
    the last ones, and when ``height`` is not a multiple of ``band`` the frame's
    last band is short (it is the last band of the rank that holds it, so the
    rank's row mapping stays first + (j // band) * stride + j % band).  Every
    rank must get at least one band."""
    if world < 1 or not (0 <= rank < world) or band < 1 or height < 1:
        raise ValueError("bad rank/world/band/height")
    bands = -(-height // band)
    if bands < world:
        raise ValueError(f"{height} rows are {bands} bands of {band} rows: fewer than {world} ranks")
    mine = (bands - rank + world - 1) // world           # bands rank, rank + world, ... below `bands`
    rows = mine * band
    if (bands - 1) % world == rank:                        # this rank holds the frame's last band, which may be short
        rows -= bands * band - height
    return rank * band, band, world * band, rows


def band_rows(height: int, world: int, rank: int, band: int = BAND_ROWS) -> list[int]:
    """Frame rows of ``rank``'s local rows 0, 1, ... under band_partition."""
    first, band, stride, rows = band_partition(height, world, rank, band)
    return [first + (j // band) * stride + j % band for j in range(rows)]


def band_share_sizes(width: int, height: int, world: int, band: int = BAND_ROWS) -> list[int]:
    """Pixels each rank holds under band_partition."""
    return [band_partition(height, world, r, band)[3] * width for r in range(world)]


def band_frame_index(width: int, height: int, world: int, band: int = BAND_ROWS):
    """Where the frame's pixels lie in the gathered [world, pad] array (pad =
    the largest share, in pixels): an int64 tensor ``idx`` of width*height
    entries such that ``gathered.reshape(-1)[idx]`` is the frame in row order."""
    import torch
    pad = max(band_share_sizes(width, height, world, band))
    src_row = torch.empty(height, dtype=torch.int64)      # frame row -> row slot in the padded gather
    for r in range(world):
        rows = torch.tensor(band_rows(height, world, r, band), dtype=torch.int64)
        src_row[rows] = r * (pad // width) + torch.arange(rows.numel(), dtype=torch.int64)
    return (src_row[:, None] * width + torch.arange(width, dtype=torch.int64)[None, :]).reshape(-1)


def gather_bands(local, width: int, height: int, dist, rank: int, world: int, dst: int = 0, band: int = BAND_ROWS):
    """Gather the ranks' interleaved rows (1-D int32 tensors of width*rows
    pixels, rows as band_partition deals them) into the full frame on ``dst``;
    None elsewhere.  Ragged shares are padded to the largest one for the gather
    (one preallocated [world, pad] tensor on ``dst``)."""
    import torch
    if world == 1:
        return local.clone()
    sizes = band_share_sizes(width, height, world, band)
    if local.numel() != sizes[rank]:
        raise ValueError(f"rank {rank} holds {local.numel()} pixels, its share is {sizes[rank]}")
    pad = max(sizes)
    buf = local
    if local.numel() != pad:
        buf = torch.zeros(pad, dtype=local.dtype, device=local.device)
        buf[:local.numel()] = local
    whole = torch.empty((world, pad), dtype=local.dtype, device=local.device) if rank == dst else None
    dist.gather(buf, list(whole.unbind(0)) if rank == dst else None, dst=dst)
    if rank != dst:
        return None
    return whole.reshape(-1)[band_frame_index(width, height, world, band).to(whole.device)]
