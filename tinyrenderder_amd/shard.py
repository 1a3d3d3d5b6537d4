"""Multi-GPU sharding of the rasterize() path: horizontal screen strips joined by ONE all-gather.

Rank r of G owns framebuffer rows [r*H/G, (r+1)*H/G): a contiguous byte range of the row-major TGAImage buffer and
of the z-buffer, so the gather is an in-place all_gather_into_tensor (RCCL over xGMI on GPUs; gloo in the CPU
tests).  Every rank streams the whole triangle list through setup (96 B/triangle of HBM reads is cheaper than
moving 128-B records over xGMI links) and rasterizes only its rows.  The reference's counters shard cleanly:
fragments_drawn adds up, the z range is a min/max, and triangle/bbox counters are identical on every rank.
"""
from __future__ import annotations

import numpy as np


def strip_rows(height: int, world: int, rank: int):
    """Rows [y0, y1) of rank `rank`.  Equal strips (the in-place all-gather needs equal chunks)."""
    if height % world:
        raise ValueError(f"height {height} is not divisible by {world} ranks: equal strips are required")
    rows = height // world
    return rank * rows, (rank + 1) * rows


def gather_strips(full, width: int, height: int, bytes_per_pixel: int, rank: int, world: int, group=None, async_op=False):
    """In-place all-gather: `full` is a flat uint8 tensor of the whole buffer whose own strip is already written.
    async_op=True returns the work handle (wait() on it before the strip is written again) instead of `full`."""
    import torch.distributed as dist
    y0, y1 = strip_rows(height, world, rank)
    chunk = full[y0 * width * bytes_per_pixel: y1 * width * bytes_per_pixel]
    work = dist.all_gather_into_tensor(full, chunk, group=group, async_op=async_op)
    return work if async_op else full


def band_rows_of(height: int, world: int, rank: int, band_rows: int):
    """Row ranges [y0, y1) of the bands rank `rank` owns under trgl_set_interleave(band_rows, rank, world)."""
    period = band_rows * world
    if height % period:
        raise ValueError(f"height {height} is not a multiple of {world} ranks x {band_rows} rows per band")
    return [(p * period + rank * band_rows, p * period + (rank + 1) * band_rows) for p in range(height // period)]


class _Works:
    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()


def gather_bands(full, width: int, height: int, bytes_per_pixel: int, band_rows: int, rank: int, world: int, group=None, async_op=False):
    """Interleaved ownership: inside each period of world * band_rows rows the bands lie in rank order, so ONE in-place
    all-gather per period (height / period of them, each over a contiguous byte range) joins the image."""
    import torch.distributed as dist
    row = width * bytes_per_pixel
    period = band_rows * world
    works = []
    for (y0, y1) in band_rows_of(height, world, rank, band_rows):
        p0 = (y0 // period) * period
        out = full[p0 * row: (p0 + period) * row]
        works.append(dist.all_gather_into_tensor(out, full[y0 * row: y1 * row], group=group, async_op=async_op))
    return _Works(works) if async_op else full


def reduce_stats(stats, device=None, group=None):
    """Combine per-rank trgl stats tuples (api.Stats.astuple()) into the whole-frame tuple.
    (The sign of a zero z-range end follows the first zero in submission order on ONE rank only; across ranks the
    merged value is the numeric min/max — DESIGN.md, multi-GPU.)"""
    import torch
    import torch.distributed as dist
    tri, frag, x0, y0, x1, y1, zlo, zhi = stats[:8]
    s = torch.tensor([frag], dtype=torch.int64, device=device)
    lo = torch.tensor([zlo], dtype=torch.float64, device=device)
    hi = torch.tensor([zhi], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    zlo, zhi = float(lo.item()), float(hi.item())
    return (tri, int(s.item()), x0, y0, x1, y1, zlo, zhi, float(np.copysign(1.0, zlo)), float(np.copysign(1.0, zhi)))


class _DevBuf:
    """Expose a raw device pointer to torch through __cuda_array_interface__."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def framebuffer_tensor(ctx):
    """A flat uint8 torch tensor that ALIASES the context's framebuffer in HBM (the in-place gather writes into it)."""
    import torch
    return torch.as_tensor(_DevBuf(ctx.framebuffer_ptr, ctx.width * ctx.height * ctx.bpp), device="cuda")


class StripLoop:
    """The per-frame sequence of ONE rank of a multi-GPU render, exactly as bench.py runs it over RCCL:

        submit (clear + draws)  ->  flush_begin (setup + binning: touches neither buffer)  ->  wait for the previous
        frame's gather  ->  flush_end (raster: writes only the rows this rank owns)  ->  start this frame's gather

    `gather()` starts joining the strips into this rank's full framebuffer and returns a handle with wait() (or None).  The
    gather of frame k therefore overlaps with the setup and binning of frame k + 1, and the rows a rank does not own keep
    the previous frame's pixels until the next gather lands."""

    def __init__(self, ctx, gather):
        # The gather is ordered against the RASTER through torch's current stream (a collective's wait() orders streams, it does
        # not block the host), so the context has to enqueue on that stream, not on the non-blocking stream it was created with.
        import torch
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        self.ctx, self.gather, self.pending = ctx, gather, None

    def step(self, submit):
        submit(self.ctx)
        self.ctx.flush_begin()
        if self.pending is not None:
            self.pending.wait()              # the raster must not overwrite the strip while the gather still reads it
        self.ctx.flush_end()
        self.pending = self.gather()

    def finish(self):
        if self.pending is not None:
            self.pending.wait()
            self.pending = None
