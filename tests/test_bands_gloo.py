"""The N > 1 path without GPUs: interleaved row-band partition + gather + un-permute, two
processes over gloo (127.0.0.1). The 'trace' is replaced by slicing a known frame, so the test
covers exactly the part that differs from single-GPU operation."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_opengl_amd import bands


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _reference_frame(h, w, dtype=torch.float32):
    g = torch.Generator().manual_seed(5)
    if dtype == torch.uint8:   # the RGBA8 target the multi-GPU bench gathers (values kept small: the payload is scaled by k+1)
        return torch.randint(0, 40, (h, w, 4), generator=g, dtype=torch.uint8)
    return torch.rand((h, w, 4), generator=g, dtype=torch.float32)


def _worker(rank, world, port, h, w, band_rows, q, dtype=torch.float32):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = _reference_frame(h, w, dtype)
        fg = bands.FrameGather(h, w, 4, band_rows, dtype, "cpu", dst=0)
        ok = True
        # frames k = 0..4 through the double-buffered protocol of bench.py: gather k is issued while
        # frame k-1 is still un-consumed; every frame carries a different payload (full * (k+1))
        pending = [None, None]
        locals_ = [fg.new_local(dtype, "cpu"), fg.new_local(dtype, "cpu")]

        def check(handle, k):
            out = fg.frame(handle)
            return torch.equal(out, full * (k + 1)) if rank == 0 else out is None

        for k in range(5):
            if pending[k & 1] is not None:
                ok = ok and check(*pending[k & 1])
            local = locals_[k & 1].zero_()
            off = 0
            for b in bands.rank_bands(h, band_rows, rank, world):
                y0, y1 = bands.band_span(h, band_rows, b)
                local[off:off + (y1 - y0)] = full[y0:y1] * (k + 1)
                off += y1 - y0
            assert off == fg.rows_local
            pending[k & 1] = (fg.gather(local, k & 1), k)
        for j in (1, 0):  # frames 3 and 4 are still pending, in that order
            ok = ok and check(*pending[j])
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("h,w,band_rows,dtype", [(64, 24, 16, torch.float32), (72, 16, 16, torch.float32), (100, 8, 8, torch.float32),
                                                  (104, 12, 8, torch.uint8)])
def test_two_rank_gather_reassembles_frame(h, w, band_rows, dtype):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, h, w, band_rows, q, dtype)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _r, ok in results)


def test_partition_covers_every_row_once():
    for h, rows, world in [(2160, 32, 8), (4320, 64, 8), (207, 8, 3), (1080, 16, 4)]:
        seen = torch.zeros(h, dtype=torch.int32)
        for r in range(world):
            for b in bands.rank_bands(h, rows, r, world):
                y0, y1 = bands.band_span(h, rows, b)
                seen[y0:y1] += 1
        assert bool((seen == 1).all())
        assert sum(bands.local_rows(h, rows, r, world) for r in range(world)) == h
        assert bands.choose_band_rows(h, world) % 8 == 0


def test_unpermute_irregular():
    h, w, rows, world = 52, 4, 8, 3
    full = _reference_frame(h, w)
    parts = []
    for r in range(world):
        buf = torch.zeros((bands.max_local_rows(h, rows, world), w, 4))
        off = 0
        for b in bands.rank_bands(h, rows, r, world):
            y0, y1 = bands.band_span(h, rows, b)
            buf[off:off + y1 - y0] = full[y0:y1]
            off += y1 - y0
        parts.append(buf)
    assert torch.equal(bands.unpermute(parts, h, rows, world), full)


# ---- the rendezvous of the one-process-per-GPU form (rtx_create_rank under torch.distributed.run; bench.py) -------------------------
def _rendezvous_worker(rank, world, port, q):
    from raytracing_opengl_amd import ranks
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def make_id():                       # stands for rtx_rccl_unique_id: must run on rank 0 ONLY
            calls.append(rank)
            return bytes((7 * k + 3) % 256 for k in range(128))

        uid = ranks.exchange_unique_id(rank, make_id)
        rays = ranks.reduce_values([1000 + rank, 3], "sum")      # per-rank ray counts -> the frame's
        worst = ranks.reduce_values([0.5 + 0.25 * rank, 2.0 - rank], "max")   # elapsed / trace time: the slowest rank's
        ranks.barrier()
        # contiguous bands weighted by kernel time (bench.py --bands balanced): every rank gathers every rank's time and computes the SAME split
        ms = ranks.gather_values(0.30 if rank == 0 else 0.10)
        split = bands.weighted_split(1080, [544, 536], ms)
        q.put((rank, uid, calls, rays, worst, ms, split))
    finally:
        dist.destroy_process_group()


def test_rank_rendezvous_hands_the_unique_id_round_and_reduces():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rendezvous_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = bytes((7 * k + 3) % 256 for k in range(128))
    assert [r[1] for r in results] == [want, want]
    assert results[0][2] == [0] and results[1][2] == []          # the id was created on rank 0 only
    assert all(r[3] == [2001.0, 6.0] and r[4] == [0.75, 2.0] for r in results)
    assert all(r[5] == [0.30, 0.10] for r in results)            # every rank's time, in rank order, on every rank
    assert results[0][6] == results[1][6] and sum(results[0][6]) == 1080 and results[0][6][0] < results[0][6][1]


def test_weighted_split_is_a_valid_split_whatever_the_times():
    """bands.weighted_split: ranges in units of 8 rows, at least one unit per rank, together the frame (the last range takes the odd rows);
    equal times keep an equal split, a slower rank gets fewer rows; and what rtx_set_band_split accepts (rtx_capi.cpp split_set)."""
    import numpy as np
    rng = np.random.default_rng(4)
    for _ in range(500):
        n = int(rng.integers(1, 9))
        h = int(rng.integers(8 * n, 5000))
        rows = bands.weighted_split(h, [h // n] * n, [1.0] * n)
        ms = list(rng.uniform(0.01, 3.0, n))
        for damping in (1.0, 0.5):
            new = bands.weighted_split(h, rows, ms, damping)
            assert sum(new) == h and len(new) == n and all(v >= 1 for v in new), (h, rows, ms, new)
            assert all(v % 8 == 0 for v in new[:-1]) and all(v >= 8 for v in new[:-1]), new
    assert bands.weighted_split(2160, [544, 536, 544, 536], [1.0, 1.0, 1.0, 1.0]) == [544, 536, 544, 536]
    slow_first = bands.weighted_split(2160, [544, 536, 544, 536], [3.0, 1.0, 1.0, 1.0])
    assert slow_first[0] < 300 and sum(slow_first) == 2160
    assert ranks_identity()


def ranks_identity():
    from raytracing_opengl_amd import ranks
    return ranks.gather_values(0.25) == [0.25]


def test_rank_rendezvous_on_one_rank_is_the_identity():
    from raytracing_opengl_amd import ranks
    assert ranks.exchange_unique_id(0, lambda: bytes(128)) == bytes(128)
    assert ranks.reduce_values([3, 4.5], "sum") == [3.0, 4.5]
    ranks.barrier()
    with pytest.raises(ValueError):
        ranks.exchange_unique_id(0, lambda: b"short")
