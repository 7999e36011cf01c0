"""End-to-end parity of the HIP engine (C ABI) with the CPU oracle: ESM-1b forward logits, and whole
Gibbs runs whose every draw is replayed by the oracle from the logits the engine emitted."""
import json
import random
import warnings

import numpy as np
import pytest

from oracle import draw as odraw
from oracle.esm_forward import EsmConfig, esm1b_forward, synthetic_esm_weights
from protein_gibbs_sampler_amd import _lib, esm_sampler, models, weights
from _standin import GOLDEN

pytestmark = pytest.mark.gpu

# Tolerance of the bf16 throughput mode on logits (fp32 residual stream, bf16 MFMA operands): stated by
# north_star as 1e-3 for the emitted logits; the bf16 mode is held to BF16_TOL here and its measured
# error is printed, see DESIGN.md "precision".
BF16_TOL = 0.15


STRICT_TOL = 1e-3    # north_star: "within 1e-3 on emitted logits" -- met by the strict precision mode (PG_PREC_FP32)


def _model(cfg_kw, sd, precision="bf16"):
    cfg = weights.make_config(weights.ESM1B_CONFIG, d_model=cfg_kw["d_model"], n_layers=cfg_kw["n_layers"],
                              d_ffn=cfg_kw["d_ffn"], max_positions=cfg_kw["max_pos"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return models.ESM1b(state_dict=sd, config=cfg, precision=precision)


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_strict_mode_logits_within_1e3(name):
    z = np.load("%s/esm_hf_%s.npz" % (GOLDEN, name))
    ck = json.loads(str(z["cfg"]))
    ocfg = EsmConfig(**ck)
    sd = synthetic_esm_weights(ocfg, seed=int(z["seed"]), std=float(z["std"]), embed_std=float(z["embed_std"]),
                               ln_jitter=float(z["ln_jitter"]))
    m = _model(ck, sd, "fp32")
    m.model.to("cuda:0")
    got = m.model.forward_logits(z["tokens"])
    want = esm1b_forward(sd, ocfg, z["tokens"])
    e_or, e_hf = np.abs(got - want).max(), np.abs(got - z["logits"]).max()
    print("\n[strict %s] max|engine - oracle| = %.3e, max|engine - HF| = %.3e (logit std %.2f)" % (name, e_or, e_hf, want.std()))
    assert e_or < STRICT_TOL and e_hf < STRICT_TOL
    assert (got.argmax(-1) == want.argmax(-1)).mean() > 0.999


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_forward_logits_vs_hf_fixture_and_oracle(name):
    z = np.load("%s/esm_hf_%s.npz" % (GOLDEN, name))
    ck = json.loads(str(z["cfg"]))
    ocfg = EsmConfig(**ck)
    sd = synthetic_esm_weights(ocfg, seed=int(z["seed"]), std=float(z["std"]), embed_std=float(z["embed_std"]),
                               ln_jitter=float(z["ln_jitter"]))
    m = _model(ck, sd)
    m.model.to("cuda:0")
    got = m.model.forward_logits(z["tokens"])
    want = esm1b_forward(sd, ocfg, z["tokens"])
    e_or, e_hf = np.abs(got - want).max(), np.abs(got - z["logits"]).max()
    print("\n[%s] max|engine - oracle| = %.3e, max|engine - HF| = %.3e, logit std = %.2f, argmax agreement = %.4f"
          % (name, e_or, e_hf, want.std(), (got.argmax(-1) == want.argmax(-1)).mean()))
    assert e_or < BF16_TOL and e_hf < BF16_TOL


def test_gibbs_run_draws_replay_exactly():
    ck = dict(d_model=128, n_layers=2, n_heads=2, d_ffn=256, max_pos=80)
    ocfg = EsmConfig(**ck)
    sd = synthetic_esm_weights(ocfg, seed=3, std=0.08, embed_std=0.5, ln_jitter=0.1)
    s = esm_sampler.ESM_sampler(_model(ck, sd), device="cuda:0")
    s.draw_seed, s.record = 99, True
    seed = "MEPAATGQEAEECAHSGRGEAWEEVMKTAYIAKQRQISFVKSHFSRQ"
    random.seed(5)
    out = s.generate(7, seed, batch_size=4, num_iters=4, num_positions_percent=20, top_k=2, burnin=2, temperature=1.1,
                     show_progress_bar=False)
    assert len(out) == 7 and all(len(x) == len(seed) for x in out)
    P = int(len(seed) * 0.2)
    random.seed(5)
    for batch_n, run in enumerate(s.last_run):
        table = np.asarray([[random.sample(range(1, len(seed) + 1), P) for _ in range(4)] for _ in range(4)])
        assert (run["table"] == table).all()                       # bit-exact positions
        tok = s.get_init_seq(seed, len(seed), 4).numpy().astype(np.int32)
        for it in range(4):
            rows = run["sampled_logits"][it].reshape(-1, 33)
            want = odraw.draw_rows(rows, s.valid_aa_idx, 2, it < 2, 1.1, batch_n * 4 + np.repeat(np.arange(4), P), it,
                                   np.tile(np.arange(P), 4), 0, 99).reshape(4, P)
            assert (want == run["sampled_tokens"][it]).all()       # bit-exact draws given the engine's logits
            # the logits the engine sampled from are the oracle's logits for the engine's token buffer
            tok_in = tok.copy()
            for b in range(4):
                tok_in[b, table[it, b]] = 32
            ref = esm1b_forward(sd, ocfg, tok_in)
            ref_rows = np.stack([ref[b, table[it, b]] for b in range(4)]).reshape(-1, 33)
            assert np.abs(rows - ref_rows).max() < BF16_TOL
            for b in range(4):
                tok[b, table[it, b]] = want[b]
        assert (tok == run["tokens"]).all()                        # write-back


def test_long_sequence_forward():
    """T > 576 tokens (ESM-1b allows 1024): online-softmax attention kernel."""
    ck = dict(d_model=128, n_layers=2, n_heads=2, d_ffn=256, max_pos=1024)
    ocfg = EsmConfig(**ck)
    sd = synthetic_esm_weights(ocfg, seed=12, std=0.08, embed_std=0.5, ln_jitter=0.1)
    m = _model(ck, sd).model.to("cuda:0")
    rng = np.random.default_rng(3)
    for T in (600, 1024):
        tok = rng.integers(4, 24, (2, T))
        tok[:, 0], tok[:, -1] = 0, 2
        tok[0, 5:40] = 32
        got = m.forward_logits(tok)
        want = esm1b_forward(sd, ocfg, tok)
        assert np.abs(got - want).max() < BF16_TOL


def test_graph_replay_matches_eager_loop():
    """Small Gibbs loops are replayed from one captured hipGraph (device-side iteration counter); the result must equal
    the eager loop (taken when per-iteration outputs are recorded) bit for bit, including burn-in and top-k switches."""
    ck = dict(d_model=128, n_layers=2, n_heads=2, d_ffn=256, max_pos=80)
    sd = synthetic_esm_weights(EsmConfig(**ck), seed=31, std=0.08, embed_std=0.5, ln_jitter=0.1)
    s = esm_sampler.ESM_sampler(_model(ck, sd), device="cuda:0")
    seed = "MEPAATGQEAEECAHSGRGEAWEEVMKTAYIAKQRQISFVKSHFSRQ"
    outs = []
    for record in (False, True, False):
        s.draw_seed, s.record = 5, record
        random.seed(9)
        outs.append(s.generate(6, seed, batch_size=3, num_iters=7, num_positions=5, top_k=2, burnin=3, temperature=0.9,
                               show_progress_bar=False))
    assert outs[0] == outs[1] == outs[2]
    assert len(set(outs[0])) > 1
    # a larger job re-allocates the workspace: the cached graph must not be replayed with stale pointers
    s.record = False
    random.seed(1)
    big = s.generate(40, seed, batch_size=40, num_iters=3, num_positions=5, show_progress_bar=False)
    assert len(big) == 40
    s.draw_seed = 5
    random.seed(9)
    again = s.generate(6, seed, batch_size=3, num_iters=7, num_positions=5, top_k=2, burnin=3, temperature=0.9, show_progress_bar=False)
    assert again == outs[0]


def test_graph_is_captured_once_and_replayed_by_later_calls():
    """BASELINE config 1's shape of use -- one generate() per sequence, a fresh draw seed, top_k and burn-in per call: the
    few-token loop captures ONE iteration once and every later call of the same shape replays it for ALL its iterations (the
    sampling parameters live in a device-side state block).  Asserts the path taken (pg_engine_get_stat) and that the replayed
    calls equal the eager loop (record=True) draw for draw."""
    ck = dict(d_model=128, n_layers=2, n_heads=2, d_ffn=256, max_pos=80)
    sd = synthetic_esm_weights(EsmConfig(**ck), seed=31, std=0.08, embed_std=0.5, ln_jitter=0.1)
    s = esm_sampler.ESM_sampler(_model(ck, sd), device="cuda:0")
    lm = s.model.model
    seed = "MEPAATGQEAEECAHSGRGEAWEEV"
    kw = dict(batch_size=1, num_iters=20, mask=True, num_positions_percent=10, show_progress_bar=False)
    variants = [dict(top_k=1, burnin=10), dict(top_k=3, burnin=5, temperature=0.7), dict(top_k=0), dict(top_k=1, burnin=0)]

    def run(record):
        outs = []
        for i, v in enumerate(variants * 2):
            s.draw_seed, s.record = 100 + i, record
            random.seed(i)
            outs.append(s.generate(1, seed, **kw, **v))
        return outs

    c0, r0 = lm.get_stat("graph_captures"), lm.get_stat("graph_replays")
    replayed = run(False)
    caps, reps = lm.get_stat("graph_captures") - c0, lm.get_stat("graph_replays") - r0
    assert caps == 1, caps                                  # one capture for eight calls with different parameters
    assert reps == 8 * 20 - 1, reps                         # every iteration but the very first (eager, sizes the buffers)
    # two-iteration calls reuse the graph too (no capture is started for fewer than three iterations, but an existing one serves)
    s.draw_seed, s.record = 7, False
    random.seed(3)
    a = s.generate(1, seed, **dict(kw, num_iters=2), top_k=1, burnin=1)
    assert lm.get_stat("graph_replays") - r0 == reps + 2 and lm.get_stat("graph_captures") - c0 == 1
    eager = run(True)                                       # recording takes the eager loop
    assert lm.get_stat("graph_replays") - r0 == reps + 2
    assert replayed == eager
    assert len({o[0] for o in replayed}) > 2
    s.draw_seed, s.record = 7, True
    random.seed(3)
    assert s.generate(1, seed, **dict(kw, num_iters=2), top_k=1, burnin=1) == a


@pytest.mark.parametrize("precision,tol", [("bf16", BF16_TOL), ("fp32", STRICT_TOL)])
def test_padded_ragged_batch_forward(precision, tol):
    """A right-padded ragged batch, as the reference's log_likelihood_batch hands to `model.model(batch)`
    (esm_sampler.py:340,355): <pad> keys are masked in attention, <pad> rows are zeroed after the embedding LayerNorm and do
    not count in the token-dropout rescale -- logits at the real positions must equal the oracle's (fair-esm semantics)."""
    ck = dict(d_model=128, n_layers=2, n_heads=2, d_ffn=256, max_pos=700)
    ocfg = EsmConfig(**ck)
    sd = synthetic_esm_weights(ocfg, seed=51, std=0.08, embed_std=0.5, ln_jitter=0.1)
    m = _model(ck, sd, precision=precision).model.to("cuda:0")
    rng = np.random.default_rng(11)
    for T, lens in ((40, (38, 17, 5, 29)), (620, (618, 100))):        # whole-sequence kernel / long-sequence kernel
        tok = np.full((len(lens), T), 1, dtype=np.int64)               # <pad> = 1
        for b, n in enumerate(lens):
            tok[b, 0] = 0
            tok[b, 1:n + 1] = rng.integers(4, 24, n)
            tok[b, n + 1] = 2
            tok[b, 3] = 32
        got = m.forward_logits(tok)
        want = esm1b_forward(sd, ocfg, tok)
        real = tok != 1
        assert np.abs(got[real] - want[real]).max() < tol
        # and the unpadded sequence alone gives the same logits (padding must be invisible)
        n = lens[1]
        alone = m.forward_logits(tok[1:2, :n + 2])
        assert np.abs(alone[0] - got[1, :n + 2]).max() < (2e-3 if precision == "fp32" else 0.05)


def test_deep_ffn_takes_split_k_path():
    """d_ffn >= 2048 with few token rows: fc2 runs as parallel K-splits + one fixed-order reduction (weight-streaming kernel
    for a single chain, 64x64 tiles for a small batch).  Logits against the oracle, run-to-run bit equality, and the
    graph-replayed Gibbs loop against the eager one."""
    ck = dict(d_model=128, n_layers=2, n_heads=2, d_ffn=2048, max_pos=80)
    ocfg = EsmConfig(**ck)
    sd = synthetic_esm_weights(ocfg, seed=41, std=0.05, embed_std=0.5, ln_jitter=0.1)
    model = _model(ck, sd)
    m = model.model.to("cuda:0")
    rng = np.random.default_rng(7)
    for B, T in ((1, 27), (20, 60)):
        tok = rng.integers(4, 24, (B, T))
        tok[:, 0], tok[:, -1] = 0, 2
        tok[0, 3:6] = 32
        got = m.forward_logits(tok)
        assert np.abs(got - esm1b_forward(sd, ocfg, tok)).max() < BF16_TOL
        assert (got == m.forward_logits(tok)).all()
    s = esm_sampler.ESM_sampler(model, device="cuda:0")
    outs = []
    for record in (False, True):
        s.draw_seed, s.record = 3, record
        random.seed(4)
        outs.append(s.generate(4, "MEPAATGQEAEECAHSGRGEAWEEV", batch_size=2, num_iters=5, num_positions=4, top_k=1, burnin=2,
                               show_progress_bar=False))
    assert outs[0] == outs[1]


def test_engine_rejects_bad_weights_and_shapes():
    ck = dict(d_model=128, n_layers=1, n_heads=2, d_ffn=256, max_pos=40)
    sd = synthetic_esm_weights(EsmConfig(**ck), seed=1)
    bad = dict(sd)
    del bad["layers.0.fc1.weight"]
    with pytest.raises(_lib.PgError, match="missing tensor 'layers.0.fc1.weight'"):
        _model(ck, bad).model.to("cuda:0")
    m = _model(ck, sd).model.to("cuda:0")
    with pytest.raises(_lib.PgError, match="position table"):
        m.forward_logits(np.zeros((1, 100), dtype=np.int32))
    assert m.forward_logits(np.zeros((0, 10), dtype=np.int32)).shape == (0, 10, 33)
