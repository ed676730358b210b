"""Row-band partition of one frame over the GPUs of a node + gather of the finished frame.

The reference is single-GPU; this is the multi-GPU step BASELINE.json's north_star adds
("partition the image over the 8 GPUs of one node as tiled rows with an RCCL gather over xGMI of
the final frame").  Pixels are independent, so the only exchange is the
final gather.  Bands are INTERLEAVED (band b belongs to rank b % world): sky rows cost ~1 ray per
pixel while object rows cost several, so contiguous slabs would be badly imbalanced.

Every rank traces its bands packed back to back (rtx_draw_bands) and the root un-permutes after
one gather.  Works on CPU tensors with the gloo backend too (tests/test_bands_gloo.py).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def num_bands(height: int, band_rows: int) -> int:
    return (height + band_rows - 1) // band_rows


def rank_bands(height: int, band_rows: int, rank: int, world: int) -> list[int]:
    return list(range(rank, num_bands(height, band_rows), world))


def band_span(height: int, band_rows: int, b: int) -> tuple[int, int]:
    y0 = b * band_rows
    return y0, min(y0 + band_rows, height)


def local_rows(height: int, band_rows: int, rank: int, world: int) -> int:
    return sum(y1 - y0 for y0, y1 in (band_span(height, band_rows, b) for b in rank_bands(height, band_rows, rank, world)))


def max_local_rows(height: int, band_rows: int, world: int) -> int:
    return max(local_rows(height, band_rows, r, world) for r in range(world))


def choose_band_rows(height: int, world: int) -> int:
    """Band height for N ranks: the kernel's tile height (8 rows). The finest interleave gives the best
    balance both in rows per rank (2160 rows / 8 ranks: 34 vs 33 bands) and in content (sky vs objects
    alternate every 8 rows); the kernel's cost does not depend on how its rows are grouped."""
    return 8


def unpermute(gathered, height: int, band_rows: int, world: int) -> torch.Tensor:
    """gathered[r]: (max_local_rows, W, C) packed bands of rank r -> (height, W, C) frame. `gathered` is a list of
    tensors or one (world, max_local_rows, W, C) tensor (FrameGather receives into one, so no stacking copy).
    All complete rounds of bands (band b of round j belongs to rank b: rows (j*world + b)*band_rows ...) move with ONE
    strided copy; the at most world-1 bands of the last, incomplete round (the final one may be short) are copied one by
    one. (A per-band loop for everything costs hundreds of tiny copies per frame -- 270 bands at 2160 rows -- which
    would dominate the frame time on the root at N = 4 or 8.)"""
    w, c = gathered[0].shape[1], gathered[0].shape[2]
    frame = torch.empty((height, w, c), dtype=gathered[0].dtype, device=gathered[0].device)
    nb = num_bands(height, band_rows)
    full_bands = height // band_rows          # bands of full height
    rounds = full_bands // world              # rounds in which every rank has a full band
    if rounds > 0:
        if isinstance(gathered, torch.Tensor):
            src = gathered[:, : rounds * band_rows]
        else:
            src = torch.stack([g[: rounds * band_rows] for g in gathered], dim=0)
        src = src.unflatten(1, (rounds, band_rows))   # (world, rounds, band_rows, W, C), a view
        frame[: rounds * world * band_rows].view(rounds, world, band_rows, w, c).copy_(src.permute(1, 0, 2, 3, 4))
    for b in range(rounds * world, nb):       # last, incomplete round
        r, j = b % world, b // world
        y0, y1 = band_span(height, band_rows, b)
        frame[y0:y1] = gathered[r][j * band_rows: j * band_rows + (y1 - y0)]
    return frame


class FrameGather:
    """Pre-allocated gather of packed row bands to rank `dst`.

    gather(local, slot) starts the collective (async) and returns a handle; frame(handle) waits and
    returns the un-permuted (H, W, C) frame on dst (None elsewhere).  `local` must hold
    max_local_rows rows (pad rows are ignored).  `slot` (0/1) selects one of two receive-buffer sets on
    the root so that the gather of frame k+1 may be issued before frame k has been un-permuted
    (double buffering: a slot may be reused only after frame() was called on its previous handle).
    On one rank it degenerates to a view of the local buffer.
    """

    def __init__(self, height: int, width: int, channels: int, band_rows: int, dtype, device, dst: int = 0, group=None):
        self.height, self.width, self.channels, self.band_rows = height, width, channels, band_rows
        self.dst, self.group = dst, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rows_max = max_local_rows(height, band_rows, self.world)
        self.rows_local = local_rows(height, band_rows, self.rank, self.world)
        self.recv = None
        self.recv_all = None
        if self.world > 1 and self.rank == dst:
            # one (world, rows, W, C) tensor per slot; the gather list are its per-rank slices (contiguous views)
            self.recv_all = [torch.empty((self.world, self.rows_max, width, channels), dtype=dtype, device=device) for _slot in range(2)]
            self.recv = [[t[r] for r in range(self.world)] for t in self.recv_all]

    def new_local(self, dtype, device) -> torch.Tensor:
        return torch.empty((self.rows_max, self.width, self.channels), dtype=dtype, device=device)

    def gather(self, local: torch.Tensor, slot: int = 0):
        if self.world == 1:
            return (None, local, slot)
        work = dist.gather(local, self.recv[slot & 1] if self.rank == self.dst else None, dst=self.dst, group=self.group, async_op=True)
        return (work, local, slot)

    def frame(self, handle):
        work, local, slot = handle
        if self.world == 1:
            return local[: self.height]   # one rank holds every band in order: the packed buffer IS the frame (a view, no copy)
        work.wait()
        if self.rank != self.dst:
            return None
        return unpermute(self.recv_all[slot & 1], self.height, self.band_rows, self.world)


def weighted_split(height: int, rows_now, ms_now, damping: float = 1.0):
    """Contiguous bands (rtx.h RTX_OPT_BAND_LAYOUT 1, rtx_set_band_split): the rows each rank should trace so that the ranks' kernel times
    come out equal, from the split in use and the kernel time each rank measured with it. rate_r = rows_r / ms_r; the new share is
    proportional to the rate (moved `damping` of the way), in units of 8 rows (the kernel's tile height), at least one unit per rank,
    the last rank taking what is left of a frame whose height is not a multiple of 8. Pure arithmetic: every rank of a per-process group
    computes the same split from the same gathered numbers (bench.py, tests/test_bands_gloo.py)."""
    n = len(rows_now)
    units = (height + 7) // 8
    if n < 1 or len(ms_now) != n or units < n:
        raise ValueError("weighted_split: need one time per rank and at least one 8-row unit per rank")
    rates = [(r / m) if (m > 0 and r > 0) else 0.0 for r, m in zip(rows_now, ms_now)]
    total = sum(rates)
    if total <= 0:
        rates, total = [1.0] * n, float(n)
    u = []
    for r, rate in zip(rows_now, rates):
        have, want = r / 8.0, units * rate / total
        u.append(max(1, int(have + damping * (want - have) + 0.5)))
    k = 0
    while sum(u) != units:          # rounding: hand the difference round, one unit at a time
        if sum(u) < units:
            u[k % n] += 1
        elif u[k % n] > 1:
            u[k % n] -= 1
        k += 1
    rows, y = [], 0
    for v in u:
        take = min(v * 8, height - y)
        rows.append(take)
        y += take
    return rows
