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


# ---- the product entry point: ESM_sampler.generate / ESM_MSA_sampler.generate over torch.distributed ranks --------------
def _fake_engine_model(msa):
    """A plug-in whose `.model` is a NativeMaskedLM subclass with the device call replaced by a deterministic function of
    (global Philox row id, iteration, slot): what the sharded generate() must reproduce for any number of ranks."""
    from protein_gibbs_sampler_amd.alphabet import Alphabet
    from protein_gibbs_sampler_amd.engine import NativeMaskedLM

    class FakeLM(NativeMaskedLM):
        def __init__(self):
            self.calls = []
            self.job_items = []

        def set_job_items(self, n):          # the samplers announce the whole batch around every shard call and reset it
            self.job_items.append(int(n))

        def eval(self):
            return self

        def to(self, device):
            return self

        def gibbs_run(self, tokens, target_idx, params, want_logits=False, want_tokens=False):
            self.calls.append(tokens.shape)
            flat = tokens.reshape(-1, tokens.shape[-1])
            idx = np.asarray(target_idx).reshape(target_idx.shape[0], -1, target_idx.shape[-1])
            for it in range(idx.shape[0]):
                for r in range(flat.shape[0]):
                    for p in range(idx.shape[2]):
                        flat[r, idx[it, r, p]] = 4 + ((params.row_id_base + r) * 7 + it * 3 + p + params.rng_seed) % 20
            return None, None

    class Plug:
        pass

    plug = Plug()
    plug.alphabet = Alphabet(True, not msa)
    plug.batch_converter = plug.alphabet.get_batch_converter(msa=msa)
    plug.model = FakeLM()
    return plug


def _generate_both(seed):
    """ESM: 2 batches of 5 chains (ragged split 3 + 2 over two ranks); MSA: 2 rounds of 3 MSAs x 2 rows."""
    import torch as _t
    _t.cuda.is_available = lambda: True
    _t.cuda.device_count = lambda: 1
    from protein_gibbs_sampler_amd import esm_msa_sampler, esm_sampler
    s = esm_sampler.ESM_sampler(_fake_engine_model(False), device="gpu")
    s.draw_seed = 11
    random.seed(seed)
    a = s.generate(9, ["MEPAATGQEAEECAHSGRGEAW", "MKPAATGQEA"], batch_size=5, num_iters=3, num_positions=4, show_progress_bar=False)
    m = esm_msa_sampler.ESM_MSA_sampler(_fake_engine_model(True), device="gpu")
    m.draw_seed = 12
    b = m.generate(11, ["MEPAATGQ", "MEP-ATGQ"], batch_size=3, num_iters=2, num_positions=3, show_progress_bar=False)
    return a, b, random.getrandbits(32), s.model.model.calls, m.model.model.calls, s.model.model.job_items, m.model.model.job_items


def _gen_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = _generate_both(seed=1 if rank == 0 else 999)        # rank 1's own seed must not matter: rank 0's RNG state is broadcast
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_generate_shards_batches_over_two_ranks():
    want_a, want_b, want_state, calls_a, calls_b, ja, jb = _generate_both(seed=1)
    assert calls_a == [(5, 24)] * 2 and calls_b == [(3, 2, 9)] * 2
    assert ja == [] and jb == []                              # one process: every call is a whole job
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gen_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        a, b, state, ca, cb, ja, jb = got[rank]
        # every shard call is bracketed by the whole batch's size (5 chains / 3 MSAs) and a reset (pg_engine_set_job_items)
        assert ja == [5, 0] * 2 and jb == [3, 0] * 2
        assert a == want_a and b == want_b                    # every rank returns the full, identical result
        assert state == want_state                            # and leaves the interpreter RNG where one process would
    assert got[0][3] == [(3, 24)] * 2 and got[1][3] == [(2, 24)] * 2          # contiguous blocks: 3 + 2 chains
    assert got[0][4] == [(2, 2, 9)] * 2 and got[1][4] == [(1, 2, 9)] * 2      # 2 + 1 MSAs
