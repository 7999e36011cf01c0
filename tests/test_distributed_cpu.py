"""N > 1 sharding logic on CPU with world_size-2 gloo: every rank derives its block of the position table from the
one CPython-exact stream, updates only its chains, and a single all-gather reproduces the single-process result."""
import os
import random
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from protein_gibbs_sampler_amd import pyrandom, sharding

B, L, P, ITERS = 10, 40, 4, 3


def _fake_iteration(tokens, idx, row_id_base, it):
    """Stand-in for the GPU iteration: a deterministic function of (global chain id, iteration, slot)."""
    for b in range(tokens.shape[0]):
        for p in range(idx.shape[1]):
            tokens[b, idx[b, p]] = 4 + ((row_id_base + b) * 7 + it * 3 + p) % 20


def _run_single():
    rng = pyrandom.NativePyRandom()
    rng.seed(0)
    table = sharding.global_position_table(rng, list(range(1, L + 1)), P, ITERS, B)
    tok = np.full((B, L + 2), 5, dtype=np.int32)
    for it in range(ITERS):
        _fake_iteration(tok, table[it], 0, it)
    return tok, table


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = pyrandom.NativePyRandom()
    rng.seed(0)
    table = sharding.global_position_table(rng, list(range(1, L + 1)), P, ITERS, B)
    lo, hi = sharding.shard_range(B, world, rank)
    mine = sharding.local_slice(table, lo, hi)
    tok = np.full((hi - lo, L + 2), 5, dtype=np.int32)
    for it in range(ITERS):
        _fake_iteration(tok, mine[it], lo, it)
    counts = [sharding.shard_range(B, world, r)[1] - sharding.shard_range(B, world, r)[0] for r in range(world)]
    full = sharding.gather_tokens(dist, torch.from_numpy(tok), counts)
    if rank == 0:
        q.put(full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_sharding_matches_single_process():
    want, table = _run_single()
    random.seed(0)
    assert table.tolist() == [[random.sample(range(1, L + 1), P) for _ in range(B)] for _ in range(ITERS)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, _free_port() if r == 0 else None, q)) for r in range(2)]
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert (got == want).all()


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 256, 257):
        for w in (1, 2, 3, 4, 8):
            spans = [sharding.shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
