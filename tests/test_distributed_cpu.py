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

        def gibbs_single_batch_run(self, tokens, mask_row, target_row, step_idx, step_sample, params_list, want_logits=False,
                                   want_tokens=False):
            self.calls.append(tokens.shape)
            for s_i in range(step_idx.shape[0]):
                for b in range(tokens.shape[0]):
                    for p, pos in enumerate(step_idx[s_i, b]):
                        if pos >= 0:
                            tokens[b, mask_row, pos] = 32
                            tokens[b, target_row, pos] = 4 + (params_list[b].rng_seed * 5 + s_i * 3 + p + int(step_sample[s_i])) % 20
            return None, None

    class Plug:
        pass

    plug = Plug()
    plug.alphabet = Alphabet(True, not msa)
    plug.batch_converter = plug.alphabet.get_batch_converter(msa=msa)
    plug.model = FakeLM()
    return plug


def _generate_both(seed, shard=True, esm_seeds=("MEPAATGQEAEECAHSGRGEAW", "MKPAATGQEA")):
    """ESM: 2 batches of 5 chains (ragged split 3 + 2 over two ranks); MSA: 2 rounds of 3 MSAs x 2 rows."""
    import torch as _t
    _t.cuda.is_available = lambda: True
    _t.cuda.device_count = lambda: 1
    from protein_gibbs_sampler_amd import esm_msa_sampler, esm_sampler
    s = esm_sampler.ESM_sampler(_fake_engine_model(False), device="gpu")
    s.draw_seed = 11
    s.shard_over_ranks = shard
    random.seed(seed)
    a = s.generate(9, list(esm_seeds), batch_size=5, num_iters=3, num_positions=4, show_progress_bar=False)
    m = esm_msa_sampler.ESM_MSA_sampler(_fake_engine_model(True), device="gpu")
    m.draw_seed = 12
    m.shard_over_ranks = shard
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


# ---- sharding is opt-in, and a sharded call checks that every rank was handed the same job ---------------------
def _optin_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # (1) default: every rank samples its OWN job (own seeds, own RNG) -- nothing is split, nothing is gathered
    own = _generate_both(seed=100 + rank, shard=False)
    # (2) opted in, but the ranks disagree on the seed sequences: every rank raises before any collective on tokens
    try:
        _generate_both(seed=1, shard=True, esm_seeds=("MEPAATGQEAEECAHSGRGEAW", "MKPAATGQEA" if rank == 0 else "MKPAATGQEC"))
        err = None
    except ValueError as e:
        err = str(e)
    q.put((rank, own, err))
    dist.barrier()
    dist.destroy_process_group()


def test_sharding_is_opt_in_and_checks_the_job():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_optin_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: (own, err) for r, own, err in (q.get(timeout=180) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        own, err = got[rank]
        want = _generate_both(seed=100 + rank, shard=False)
        assert own[0] == want[0] and own[1] == want[1] and own[2] == want[2]
        assert own[3] == [(5, 24)] * 2 and own[4] == [(3, 2, 9)] * 2 and own[5] == [] and own[6] == []     # whole batches on every rank
        assert err is not None and "different arguments" in err
    assert got[0][0][0] != got[1][0][0]          # the two ranks really ran different jobs


# ---- row e5: generate_single over a list of templates: batched == serial, 2 ranks == 1 process ------------------------------
_TEMPLATES = [["MEPAATGQ", "MEP-ATGQ", "MKPAATGQ"], ["MEPAATGQ", "MEP-ATGQ", "MKPAATGQ"], ["ACDEFGHIKLMN", "ACDEFGHIKLMN"],
              ["MEPAATGA", "MEP-ATGC", "MKPAATGD"], ["ACD", "ACE"]]
_EXCL = [None, [0, 3], [], [1], None]


def _single_batch(seed, shard, max_batch=2, tseed=5):
    import torch as _t
    _t.cuda.is_available = lambda: True
    _t.cuda.device_count = lambda: 1
    from protein_gibbs_sampler_amd import esm_msa_sampler
    m = esm_msa_sampler.ESM_MSA_sampler(_fake_engine_model(True), device="gpu")
    m.shard_over_ranks = shard
    random.seed(seed)
    _t.manual_seed(tseed)
    out = m.generate_single_batch(_TEMPLATES, steps=3, passes=2, burn_in=1, target_index=-1, k=1, exclude_positions=_EXCL,
                                  max_batch=max_batch)
    return out, random.getrandbits(32), m.model.model.calls


def _single_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q.put((rank, _single_batch(seed=3 if rank == 0 else 1234, shard=True, tseed=5 if rank == 0 else 99)))
    dist.barrier()
    dist.destroy_process_group()


def test_generate_single_batch_equals_serial_calls_and_shards_over_two_ranks():
    import torch as _t
    _t.cuda.is_available = lambda: True
    _t.cuda.device_count = lambda: 1
    from protein_gibbs_sampler_amd import esm_msa_sampler
    m = esm_msa_sampler.ESM_MSA_sampler(_fake_engine_model(True), device="gpu")
    random.seed(3)
    _t.manual_seed(5)
    serial = [m.generate_single(t, steps=3, passes=2, burn_in=1, target_index=-1, k=1, exclude_positions=e)
              for t, e in zip(_TEMPLATES, _EXCL)]
    state = random.getrandbits(32)
    batched, bstate, calls = _single_batch(seed=3, shard=False)
    assert batched == serial and bstate == state
    # equal-shape templates share a call (templates 0, 1, 3 are 3 x 9; max_batch = 2 -> 2 + 1), the others go alone
    assert sorted(calls) == sorted([(2, 3, 9), (1, 3, 9), (1, 2, 13), (1, 2, 4)])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_single_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        out, st, _ = got[rank]
        assert out == serial and st == state          # every rank returns the whole list; rank 0's RNG state and torch seeds rule
    assert sorted(got[0][2]) == sorted([(2, 3, 9), (1, 2, 13)])      # rank 0: templates 0-2
    assert sorted(got[1][2]) == sorted([(1, 3, 9), (1, 2, 4)])       # rank 1: templates 3-4
