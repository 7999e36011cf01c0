"""N > 1 rehearsal of the job's one collective without a multi-GPU node (the buffers gathered are what
/root/reference/src/pgen/esm_sampler.py:236-239 untokenises).

RCCL has only ever run at world size 1 on the builder's boxes, so the part of `pg_gather_tokens` that is NOT RCCL -- which form
is taken, the block size, the scratch layout, the per-rank pack offsets (csrc/comm.cpp: GatherPlan + gather_with) -- is executed
here at world sizes 2 ... 8 through `pg_dbg_gather_tokens_host`: the same code on host buffers with the all-gather injected from
the test (one thread per simulated rank, a barrier-synchronised exchange).  The padded branch of `sharding.gather_tokens`
(torch.distributed form) gets the same treatment through a thread-backed stand-in for the `dist` module.  Shard shapes: equal,
ragged, ranks with zero rows, everything empty."""
import ctypes
import threading

import numpy as np
import pytest

from protein_gibbs_sampler_amd import _lib, sharding

AG_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)

COUNT_CASES = [
    [3, 3], [3, 2], [0, 4], [4, 0], [0, 0],
    [2, 2, 2], [3, 2, 2], [0, 5, 1], [1, 0, 0],
    [2, 2, 2, 2], [2, 2, 1, 1], [0, 0, 3, 0],
    [5, 4, 4, 4, 4], [1, 1, 1, 1, 1, 1], [3, 3, 2, 2, 2, 2], [0, 1, 2, 3, 4, 5, 6],
    [32] * 8, [33] * 4 + [32] * 4, [1, 0, 1, 0, 1, 0, 1, 0], [0] * 8, [0, 0, 0, 0, 0, 0, 0, 7],
]


def _rows(rank, n, width):
    """rank's rows: value encodes (rank, row, column) so that a misplaced or padded row is visible."""
    r = np.arange(n, dtype=np.int64)[:, None] * 1000 + np.arange(width, dtype=np.int64)[None, :]
    return (rank * 1000000 + r + 1).astype(np.int32).reshape(n, width)


class ThreadExchange:
    """An all-gather among `world` threads: everybody deposits its block, waits, reads all blocks in rank order."""

    def __init__(self, world):
        self.world, self.slots = world, [None] * world
        self.barrier = threading.Barrier(world)
        self.calls = [0] * world

    def all_gather(self, rank, block):
        self.slots[rank] = block
        self.barrier.wait(timeout=30)
        got = [self.slots[r] for r in range(self.world)]
        self.barrier.wait(timeout=30)           # nobody overwrites a slot before everybody has read it
        self.calls[rank] += 1
        return got


def _run_ranks(world, fn):
    out, errs = [None] * world, []

    def body(r):
        try:
            out[r] = fn(r)
        except BaseException as e:            # noqa: BLE001 -- reported below, with the rank
            errs.append((r, e))
    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(60) for t in ts]
    assert not errs, errs
    return out


@pytest.mark.parametrize("force_padded", [False, True])
@pytest.mark.parametrize("counts", COUNT_CASES, ids=lambda c: "-".join(map(str, c)))
def test_c_abi_gather_bookkeeping_at_world_2_to_8(counts, force_padded):
    L = _lib.lib()
    world, width = len(counts), 7
    ex = ThreadExchange(world)
    want = np.concatenate([_rows(r, counts[r], width) for r in range(world)]) if sum(counts) else np.zeros((0, width), np.int32)

    def rank_body(rank):
        @AG_FN
        def ag(_ctx, send, recv, n):
            mine = np.ctypeslib.as_array(ctypes.cast(send, ctypes.POINTER(ctypes.c_int32)), shape=(n,)).copy()
            blocks = ex.all_gather(rank, mine)
            assert all(len(b) == n for b in blocks), "ranks disagree about the block size"
            dst = np.ctypeslib.as_array(ctypes.cast(recv, ctypes.POINTER(ctypes.c_int32)), shape=(n * world,))
            dst[:] = np.concatenate(blocks)
            return 0
        local = np.ascontiguousarray(_rows(rank, counts[rank], width))
        out = np.full((sum(counts) + 1, width), -7, np.int32)              # one guard row behind the output
        c_counts = (ctypes.c_int64 * world)(*counts)
        _lib.check(L.pg_dbg_gather_tokens_host(rank, world, _lib.ptr(local) if counts[rank] else None, counts[rank], width, c_counts,
                                               1 if force_padded else 0, ctypes.cast(ag, ctypes.c_void_p), None, _lib.ptr(out)))
        return out

    outs = _run_ranks(world, rank_body)
    for rank, out in enumerate(outs):
        assert (out[:-1] == want).all(), "rank %d got a different gathered buffer" % rank
        assert (out[-1] == -7).all(), "rank %d wrote behind the output" % rank
    # the collective is entered by every rank or by none (a rank that skipped it would leave the others hanging on RCCL)
    assert len(set(ex.calls)) == 1
    assert ex.calls[0] == (0 if sum(counts) == 0 or (force_padded and max(counts) == 0) else 1)


@pytest.mark.parametrize("counts", COUNT_CASES, ids=lambda c: "-".join(map(str, c)))
def test_gather_plan_layout(counts):
    """The plan itself: form, block size, scratch size and pack offsets, identical on every rank."""
    L = _lib.lib()
    world, width = len(counts), 258
    c_counts = (ctypes.c_int64 * world)(*counts)
    plans = []
    for rank in range(world):
        o7 = (ctypes.c_int64 * 7)()
        offs = (ctypes.c_int64 * world)()
        _lib.check(L.pg_dbg_gather_plan(rank, world, counts[rank], width, c_counts, 0, o7, offs))
        plans.append((list(o7), list(offs)))
    equal = len(set(counts)) == 1
    mx = max(counts)
    for rank, (o7, offs) in enumerate(plans):
        assert o7[0] == int(equal)
        assert o7[1] == int(mx == 0)
        assert o7[3] == width * 4 and o7[6] == sum(counts) * width * 4
        if not equal:
            assert o7[2] == mx and o7[4] == mx * width * 4 and o7[5] == (world + 1) * mx * width * 4
        assert offs == [sum(counts[:r]) * width * 4 for r in range(world)]
    assert all(p[0][:2] == plans[0][0][:2] for p in plans), "ranks would take different forms of the collective"


def test_gather_rejects_inconsistent_arguments():
    L = _lib.lib()
    out = np.zeros((4, 3), np.int32)
    loc = np.zeros((2, 3), np.int32)

    @AG_FN
    def ag(_ctx, send, recv, n):
        return 0
    agp = ctypes.cast(ag, ctypes.c_void_p)
    bad = (ctypes.c_int64 * 2)(1, 2)          # counts[rank] != rows
    assert L.pg_dbg_gather_tokens_host(0, 2, _lib.ptr(loc), 2, 3, bad, 0, agp, None, _lib.ptr(out)) == _lib.PG_ERR_INVALID
    neg = (ctypes.c_int64 * 2)(2, -1)
    assert L.pg_dbg_gather_tokens_host(0, 2, _lib.ptr(loc), 2, 3, neg, 0, agp, None, _lib.ptr(out)) == _lib.PG_ERR_INVALID
    ok = (ctypes.c_int64 * 2)(2, 2)
    assert L.pg_dbg_gather_tokens_host(2, 2, _lib.ptr(loc), 2, 3, ok, 0, agp, None, _lib.ptr(out)) == _lib.PG_ERR_INVALID   # rank == world
    assert L.pg_dbg_gather_tokens_host(0, 2, None, 2, 3, ok, 0, agp, None, _lib.ptr(out)) == _lib.PG_ERR_INVALID            # rows without a buffer

    @AG_FN
    def failing(_ctx, send, recv, n):
        return 5
    assert L.pg_dbg_gather_tokens_host(0, 2, _lib.ptr(loc), 2, 3, ok, 0, ctypes.cast(failing, ctypes.c_void_p), None,
                                       _lib.ptr(out)) == _lib.PG_ERR_HIP


class ThreadDist:
    """What sharding.gather_tokens needs of torch.distributed, backed by ThreadExchange: per-thread rank, gloo-style all_gather."""

    def __init__(self, world):
        self.ex = ThreadExchange(world)
        self.local = threading.local()

    def get_world_size(self):
        return self.ex.world

    def get_backend(self):
        return "gloo"

    def all_gather(self, parts, t):
        import torch
        got = self.ex.all_gather(self.local.rank, t.clone())
        assert all(g.shape == t.shape for g in got), "ranks entered all_gather with different block shapes"
        for p, g in zip(parts, got):
            p.copy_(torch.as_tensor(g))


@pytest.mark.parametrize("force_padded", [False, True])
@pytest.mark.parametrize("counts", COUNT_CASES, ids=lambda c: "-".join(map(str, c)))
def test_sharding_gather_tokens_at_world_2_to_8(counts, force_padded):
    import torch
    world, width = len(counts), 5
    d = ThreadDist(world)
    want = np.concatenate([_rows(r, counts[r], width) for r in range(world)]) if sum(counts) else np.zeros((0, width), np.int32)

    def rank_body(rank):
        d.local.rank = rank
        return sharding.gather_tokens(d, torch.from_numpy(_rows(rank, counts[rank], width)), counts, force_padded=force_padded).numpy()

    for rank, out in enumerate(_run_ranks(world, rank_body)):
        assert out.shape == want.shape and (out == want).all(), "rank %d" % rank
    assert len(set(d.ex.calls)) == 1 and d.ex.calls[0] == 1


def test_shard_ranges_cover_every_job_size():
    for world in range(1, 9):
        for n in (0, 1, 7, 8, 31, 32, 255, 256, 257):
            blocks = [sharding.shard_range(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
