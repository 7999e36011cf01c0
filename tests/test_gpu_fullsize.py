"""BASELINE configuration 2 at FULL size (ESM-1b: 33 layers, d = 1280; 256 chains x L = 256) on the GPU, checked through
size-independent properties -- the CPU oracle needs minutes per chain at this size:
  * position selection equals CPython's `random.sample` stream for all 256 x 25 x iterations draws,
  * every draw, replayed by the oracle from the 33 logits the engine sampled from, gives the identical token (pg_draw v1 is
    bit-exact), only the selected positions change, they end up holding valid residues, nothing else is touched,
  * the run is bit-reproducible, and a chain's result does not depend on which other chains share the batch's GEMM tiles
    beyond the logits' accumulation noise (row-wise independence: logits of a sub-batch vs the full batch),
  * forward logits are finite and invariant to a re-run.
"""
import random
import warnings

import numpy as np
import pytest

from oracle import draw as odraw
from protein_gibbs_sampler_amd import esm_sampler, models, weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sampler():
    cfg = dict(weights.ESM1B_CONFIG)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = models.ESM1b(state_dict=weights.synthetic_state_dict(cfg, seed=0), config=cfg)
    return esm_sampler.ESM_sampler(m, device="cuda:0")


def _seeds(n, L):
    rng = np.random.default_rng(1234)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    return ["".join(aa[i] for i in row) for row in rng.integers(0, 20, (n, L))]


def test_config2_full_size_gibbs_properties(sampler):
    B, L, P, iters = 256, 256, 25, 2
    seeds = _seeds(B, L)
    s = sampler
    outs, runs = [], []
    for _ in range(2):
        s.draw_seed, s.record = 7, True
        random.seed(0)
        # a list seed_seq draws one seed per chain with random.choices (esm_sampler.py:112): identical in both runs
        outs.append(s.generate(B, seeds, batch_size=B, num_iters=iters, num_positions_percent=10, top_k=0, temperature=1.0,
                               burnin=float("inf"), show_progress_bar=False))
        runs.append(s.last_run[0])
    assert outs[0] == outs[1]                                          # bit-reproducible
    run = runs[0]
    random.seed(0)
    chosen = random.choices(seeds, k=B)
    table = np.asarray([[random.sample(range(1, L + 1), P) for _ in range(B)] for _ in range(iters)])
    assert (run["table"] == table).all()                               # 12 800 position draws, bit-exact
    alpha = s.model.alphabet
    tok = np.asarray([[alpha.cls_idx] + [alpha.get_idx(c) for c in x] + [alpha.eos_idx] for x in chosen], dtype=np.int32)
    start = tok.copy()
    for it in range(iters):
        rows = run["sampled_logits"][it].reshape(-1, 33)
        assert np.isfinite(rows).all()
        want = odraw.draw_rows(rows, s.valid_aa_idx, 0, True, 1.0, np.repeat(np.arange(B), P), it, np.tile(np.arange(P), B), 0, 7)
        assert (want.reshape(B, P) == run["sampled_tokens"][it]).all()  # 6400 draws per iteration, bit-exact given the logits
        for b in range(B):
            tok[b, table[it, b]] = want.reshape(B, P)[b]
    assert (tok == run["tokens"]).all()                                # write-back, nothing else touched
    touched = np.zeros_like(tok, dtype=bool)
    for it in range(iters):
        for b in range(B):
            touched[b, table[it, b]] = True
    assert (tok[~touched] == start[~touched]).all()
    assert np.isin(tok[touched], s.valid_aa_idx).all() and not (tok == 32).any()
    assert all(len(x) == L for x in outs[0])


def test_config2_full_size_rows_are_independent(sampler):
    """Chains never interact, and every large-batch GEMM kernel (256x256 ping-pong tiles, the peeled panels' 64x64 / 128x128
    tiles) accumulates a row's dot products in the same order: the logits of a chain are BIT-identical whether it runs in
    the full 256-chain batch or in a 128- or 100-chain shard (this is what makes multi-GPU output independent of the
    sharding).  A handful of chains alone take the split-K path for fc2: equal up to bf16 rounding flips."""
    B, L = 256, 256
    s = sampler
    rng = np.random.default_rng(5)
    tok = np.concatenate([np.zeros((B, 1), np.int64), rng.integers(4, 24, (B, L)), np.full((B, 1), 2)], axis=1)
    tok[:, 10:40:3] = 32
    lm = s.model.model
    full = lm.forward_logits(tok)
    assert np.isfinite(full).all() and (full == lm.forward_logits(tok)).all()
    assert (lm.forward_logits(tok[:128]) == full[:128]).all()
    assert (lm.forward_logits(tok[128:]) == full[128:]).all()         # includes the rows of the full batch's peeled panels
    assert (lm.forward_logits(tok[100:200]) == full[100:200]).all()
    # BASELINE config 3 gives each of 8 GPUs a 32-chain shard (8256 token rows): same logits, bit for bit, and 64 / 128 for
    # the 4- and 2-GPU points of the scaling curve
    for g in (0, 3, 7):
        assert (lm.forward_logits(tok[g * 32:(g + 1) * 32]) == full[g * 32:(g + 1) * 32]).all()
    assert (lm.forward_logits(tok[64:128]) == full[64:128]).all()
    few = lm.forward_logits(tok[-3:])
    d = np.abs(few - full[-3:]).max()
    print("\nfull-size: 3 chains alone vs in the batch: max|diff| = %.3e (logit std %.2f)" % (d, full.std()))
    assert d < 0.05


def _sharded_gibbs(sampler, B, P, iters, worlds, job_items=False):
    """Run the B-chain job whole and as `world` contiguous shards (one after the other on this GPU, each with its slice of the
    one position stream and its global chain ids); returns nothing, asserts tokens AND the logits of every draw bit for bit."""
    import ctypes
    import torch
    from protein_gibbs_sampler_amd import _lib, pyrandom, sharding
    L = 256
    T = L + 2
    rng = np.random.default_rng(1234)
    tok_all = np.concatenate([np.zeros((B, 1), np.int64), rng.integers(4, 24, (B, L)), np.full((B, 1), 2)], axis=1).astype(np.int32)
    lm = sampler.model.model
    L_ = _lib.lib()

    def run(lo, hi):
        r = pyrandom.NativePyRandom()
        r.seed(0)
        table = sharding.local_slice(sharding.global_position_table(r, list(range(1, L + 1)), P, iters, B), lo, hi)
        params = _lib.make_sample_params(True, 32, 0, float("inf"), 1.0, sampler.valid_aa_idx, rng_seed=0, row_id_base=lo)
        d_tok = torch.from_numpy(tok_all[lo:hi].copy()).cuda()
        d_idx = torch.from_numpy(table).cuda()
        d_lg = torch.empty((iters, hi - lo, P, 33), dtype=torch.float32, device="cuda")
        if job_items:
            lm.set_job_items(B)              # as the samplers and bench.py do on every rank (pgibbs.h pg_engine_set_job_items)
        try:
            _lib.check(L_.pg_esm_gibbs_run_device(lm.handle, ctypes.c_void_p(d_tok.data_ptr()), hi - lo, T,
                                                  ctypes.c_void_p(d_idx.data_ptr()), iters, P, ctypes.byref(params),
                                                  ctypes.c_void_p(d_lg.data_ptr()), None))
            lm.synchronize()
        finally:
            lm.set_job_items(0)
        return d_tok.cpu().numpy(), d_lg.cpu().numpy()

    whole, whole_lg = run(0, B)
    assert (whole != tok_all).any()
    for world in worlds:
        parts = [run(*sharding.shard_range(B, world, g)) for g in range(world)]
        # the logits every draw was made from, bit for bit (equal tokens alone could be luck: a 1e-3 logit difference -- a K-split
        # fc2 on a shard's 800 selected rows did that until round 2 -- flips only one draw in a few thousand)
        assert (np.concatenate([p[1] for p in parts], axis=1) == whole_lg).all(), "B=%d P=%d world=%d: sampled-position logits differ" % (B, P, world)
        assert (np.concatenate([p[0] for p in parts]) == whole).all(), "B=%d P=%d world=%d" % (B, P, world)


def test_config3_shards_reproduce_the_single_gpu_gibbs_run(sampler):
    """The whole Gibbs job of config 2, re-run as the 32- / 64- / 128-chain shards config 3 puts on 8 / 4 / 2 GPUs."""
    _sharded_gibbs(sampler, 256, 25, 3, (8, 4, 2))


@pytest.mark.parametrize("B,P", [(64, 1), (64, 5), (128, 3)])
def test_shards_with_few_selected_rows(sampler, B, P):
    """Shards whose B*P selected rows are few (8 to 48: the shapes that would pick the weight-streaming GEMM or a K-split for the
    pruned last layer and the LM head) against the whole batch (64 to 384 selected rows: tile kernels).  The contract
    (DESIGN.md section 7): without being told the job's size, bit-identical for any contiguous split whose shards each hold
    more than 2048 token rows."""
    _sharded_gibbs(sampler, B, P, 2, (8,))


@pytest.mark.parametrize("B,P,world", [(24, 2, 8), (16, 25, 16), (256, 25, 8)])
def test_shards_that_know_the_job_size(sampler, B, P, world):
    """With pg_engine_set_job_items (what ESM_sampler.generate and bench.py do on every rank) every contiguous split of a job
    with more than 2048 token rows is bit-identical with the single-engine run -- also three chains or one chain per shard,
    where a shard on its own would pick the few-chain kernels (weight streaming, K-splits) and land 1e-2 away."""
    _sharded_gibbs(sampler, B, P, 2, (world,), job_items=True)


def test_config4_full_size_msa_gibbs_properties():
    """BASELINE configuration 4 at full size (ESM-MSA-1b: 12 layers, d = 768; 64 MSAs x depth 32 x L = 256, 25 positions per
    row): 51 200 position draws equal CPython's stream, every draw replays bit-exactly from the engine's logits, only the
    selected positions change, and an MSA's logits do not depend on how many other MSAs share the batch."""
    from protein_gibbs_sampler_amd import esm_msa_sampler
    cfg = dict(weights.MSA1B_CONFIG)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = models.ESM_MSA1(state_dict=weights.synthetic_state_dict(cfg, seed=0), config=cfg)
    s = esm_msa_sampler.ESM_MSA_sampler(m, device="cuda:0")
    B, R, L, P = 64, 32, 256, 25
    rng = np.random.default_rng(1234)
    sym = np.asarray(list("ACDEFGHIKLMNPQRSTVWY"))
    rows = sym[rng.integers(0, 20, (R, L))]
    rows[rng.random((R, L)) < 0.1] = "-"
    msa = ["".join(r) for r in rows]
    s.draw_seed, s.record = 3, True
    random.seed(0)
    out = s.generate(B * R, msa, batch_size=B, num_iters=1, num_positions=P, top_k=0, temperature=1.0, burnin=float("inf"),
                     show_progress_bar=False)
    assert len(out) == B * R and all(len(x) == L for x in out)
    run = s.last_run[0]
    random.seed(0)
    table = np.asarray([[[random.sample(range(1, L + 1), P) for _ in range(R)] for _ in range(B)]])
    assert (run["table"] == table).all()
    lg = run["sampled_logits"][0].reshape(-1, 33)
    assert np.isfinite(lg).all()
    want = odraw.draw_rows(lg, s.valid_aa_idx, 0, True, 1.0, np.repeat(np.arange(B * R), P), 0, np.tile(np.arange(P), B * R), 0, 3)
    assert (want.reshape(B, R, P) == run["sampled_tokens"][0]).all()
    tok = s.get_init_msa(msa, L, B).numpy().astype(np.int32)
    start = tok.copy()
    bi, ri = np.meshgrid(np.arange(B), np.arange(R), indexing="ij")
    tok[bi[..., None], ri[..., None], table[0]] = want.reshape(B, R, P)
    assert (tok == run["tokens"]).all()
    touched = np.zeros_like(tok, dtype=bool)
    touched[bi[..., None], ri[..., None], table[0]] = True
    assert (tok[~touched] == start[~touched]).all() and np.isin(tok[touched], s.valid_aa_idx).all()
    lm = m.model
    full = lm.forward_logits(start[:16])
    assert (lm.forward_logits(start[:8]) == full[:8]).all() and (lm.forward_logits(start[8:16]) == full[8:16]).all()


def test_strict_mode_shards_reproduce_the_single_gpu_gibbs_run():
    """The strict precision mode (PG_PREC_FP32), 64 chains whole vs 8 shards of 8: the whole batch runs every projection on the fused
    three-product kernel, an 8-chain shard takes the plain GEMM over K' = 3K for its 45-tile out-proj / fc2 -- same k order, same
    bits, so the logits of every draw and the tokens are identical."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wrapper = models.ESM1b(state_dict=weights.synthetic_state_dict(dict(weights.ESM1B_CONFIG), seed=0),
                               config=dict(weights.ESM1B_CONFIG), precision="fp32")
    s = esm_sampler.ESM_sampler(wrapper, device="cuda:0")
    _sharded_gibbs(s, 64, 25, 2, (8,), job_items=True)


@pytest.mark.parametrize("B,R,L,P,worlds", [(8, 8, 100, 3, (8, 2)), (4, 16, 256, 5, (4,)), (16, 32, 256, 25, (16,))])
def test_msa_shards_reproduce_the_single_gpu_gibbs_run(B, R, L, P, worlds):
    """ESM-MSA-1b at full size: B MSAs whole vs contiguous shards that know the job's size -- tokens and the logits of every draw
    bit for bit.  One or two MSAs per shard on their own would take the row-split form of the tied row attention (sum over
    alignment rows in chunks) where the whole batch does not; the decision is taken on the job's batch."""
    import ctypes
    import torch
    from protein_gibbs_sampler_amd import _lib, pyrandom, sharding
    cfg = dict(weights.MSA1B_CONFIG)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wrapper = models.ESM_MSA1(state_dict=weights.synthetic_state_dict(cfg, seed=0), config=cfg)
    lm = wrapper.model.to("cuda:0")
    valid = sorted(wrapper.alphabet.get_idx(t) for t in "-ACDEFGHIKLMNPQRSTVWY")
    rng = np.random.default_rng(1234)
    C, iters = L + 1, 2
    aa = np.asarray(valid[:20])[rng.integers(0, 20, (B, R, L))]
    tok_all = np.concatenate([np.zeros((B, R, 1), np.int64), aa], axis=2).astype(np.int32)
    L_ = _lib.lib()

    def run(lo, hi):
        r = pyrandom.NativePyRandom()
        r.seed(0)
        table = r.sample_table(list(range(1, L + 1)), P, iters * B * R).reshape(iters, B, R, P)[:, lo:hi].copy()
        params = _lib.make_sample_params(True, cfg["mask_idx"], 0, float("inf"), 1.0, valid, rng_seed=0, row_id_base=lo * R)
        d_tok = torch.from_numpy(tok_all[lo:hi].copy()).cuda()
        d_idx = torch.from_numpy(table).cuda()
        d_lg = torch.empty((iters, hi - lo, R, P, 33), dtype=torch.float32, device="cuda")
        lm.set_job_items(B)
        try:
            _lib.check(L_.pg_msa_gibbs_run_device(lm.handle, ctypes.c_void_p(d_tok.data_ptr()), hi - lo, R, C,
                                                  ctypes.c_void_p(d_idx.data_ptr()), iters, P, ctypes.byref(params),
                                                  ctypes.c_void_p(d_lg.data_ptr()), None))
            lm.synchronize()
        finally:
            lm.set_job_items(0)
        return d_tok.cpu().numpy(), d_lg.cpu().numpy()

    whole, whole_lg = run(0, B)
    assert (whole != tok_all).any()
    for world in worlds:
        parts = [run(*sharding.shard_range(B, world, g)) for g in range(world)]
        assert (np.concatenate([p[1] for p in parts], axis=1) == whole_lg).all(), "world=%d: sampled-position logits differ" % world
        assert (np.concatenate([p[0] for p in parts]) == whole).all(), "world=%d" % world
