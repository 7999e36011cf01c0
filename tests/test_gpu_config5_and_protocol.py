"""BASELINE configuration 5 on the full ESM-MSA-1b engine under pytest, and the reference's plug-in call protocol
(`self.model.model(batch)["logits"]`) driving the HIP engine.

config 5 = `pgen_msa_revised` single-row resample: one template MSA of depth 128 x L = 512 (C = 513 columns),
steps = 10, passes = 3, burn_in = 2, k = 1 (/root/reference/src/pgen/esm_msa_sampler.py:101-147).  One forward is
16.5 TFLOP, i.e. minutes on the CPU oracle, so the 30 forwards are checked through size-independent properties:
the step lists equal the reference's `random.shuffle` + `partition` stream, every draw replays bit-exactly through
oracle.draw from the logits the engine emitted, only row `target_index` receives tokens, row -1 is the one masked
(quirk Q2, :133), and one forward's logits at this shape are compared with the fp32 oracle restricted to what is
cheap: R == 1 rows through the oracle are covered elsewhere; here the first step's logits must equal
`forward_logits` of the same masked token buffer (the pruned last layer + LM-head-at-sampled-rows path against
the unpruned all-rows path).
"""
import random
import warnings

import numpy as np
import pytest
import torch

from oracle import draw as odraw
from protein_gibbs_sampler_amd import esm_msa_sampler, esm_sampler, models, weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def msa_model():
    cfg = dict(weights.MSA1B_CONFIG)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return models.ESM_MSA1(state_dict=weights.synthetic_state_dict(cfg, seed=2, std=0.025, embed_std=0.3, ln_jitter=0.1),
                               config=cfg)


def _template_msa(R, L, seed=1234):
    rng = np.random.default_rng(seed)
    sym = np.asarray(list("ACDEFGHIKLMNPQRSTVWY"))
    rows = sym[rng.integers(0, 20, (R, L))]
    rows[rng.random((R, L)) < 0.1] = "-"
    return ["".join(r) for r in rows]


@pytest.mark.parametrize("target_index", [0, -1])
def test_config5_generate_single_full_size(msa_model, target_index):
    R, L, steps, passes, burn_in = 128, 512, 10, 3, 2
    s = esm_msa_sampler.ESM_MSA_sampler(msa_model, device="cuda:0")
    msa = _template_msa(R, L)
    s.draw_seed, s.record = 17, True
    random.seed(5)
    out = s.generate_single(list(msa), steps=steps, passes=passes, burn_in=burn_in, target_index=target_index, k=1)
    run = s.last_run[0]
    assert len(out) == L and set(out) <= set(esm_msa_sampler.ESM_MSA_ALLOWED_AMINO_ACIDS)

    # positions: random.shuffle + partition, exactly the reference's consumption of the interpreter's RNG
    random.seed(5)
    positions = list(range(1, L + 1))
    step_lists = []
    for _ in range(passes):
        random.shuffle(positions)
        step_lists += esm_msa_sampler.partition(positions, steps)
    assert len(step_lists) == steps * passes == run["table"].shape[0]
    assert sorted(len(x) for x in step_lists[:steps]) == [51] * 8 + [52] * 2          # 512 = 2*52 + 8*51
    assert random.getrandbits(32) == _next_after(5, passes, L)

    tr = target_index % R
    tok = s.get_init_msa(msa, L, 1).numpy().astype(np.int32)
    start = tok.copy()
    for i, st in enumerate(step_lists):
        n = len(st)
        assert run["table"][i, 0, :n].tolist() == st and (run["table"][i, 0, n:] == -1).all()
        rows = run["sampled_logits"][i][:n]
        assert np.isfinite(rows).all()
        # sample=(pass_num < burn_in): full-distribution draw during burn-in, argmax (k = 1) afterwards
        want = odraw.draw_rows(rows, s.valid_aa_idx, 1, i < burn_in * steps, None, [tr] * n, i, np.arange(n), 0, 17)
        assert (want == run["sampled_tokens"][i][:n]).all()
        if i >= burn_in * steps:                                                        # argmax over the 20 residues
            assert (want == np.asarray(s.valid_aa_idx)[rows[:, s.valid_aa_idx].argmax(-1)]).all()
        tok[0, R - 1, st] = 32                                                          # row -1 is the one masked
        tok[0, tr, st] = want                                                           # row target_index is sampled
    assert (tok == run["tokens"]).all()
    untouched = np.ones(R, bool)
    untouched[[tr, R - 1]] = False
    assert (run["tokens"][0, untouched] == start[0, untouched]).all()
    if tr != R - 1:
        assert (run["tokens"][0, R - 1, 1:] == 32).all()                                # never un-masked (quirk Q2)
    assert out == s.untokenize_batch(torch.from_numpy(tok.astype(np.int64)))[target_index]


def _next_after(seed, passes, L):
    random.seed(seed)
    p = list(range(1, L + 1))
    for _ in range(passes):
        random.shuffle(p)
    return random.getrandbits(32)


def test_config5_first_step_logits_equal_unpruned_forward(msa_model):
    """The Gibbs path prunes the last layer to the sampled rows and evaluates the LM head only there; the logits it samples
    from must be the logits `model.model(batch)["logits"]` gives at those positions (all rows, unpruned)."""
    R, L = 128, 512
    s = esm_msa_sampler.ESM_MSA_sampler(msa_model, device="cuda:0")
    msa = _template_msa(R, L, seed=77)
    s.draw_seed, s.record = 3, True
    random.seed(9)
    s.generate_single(list(msa), steps=10, passes=1, burn_in=1, target_index=4, k=1)
    run = s.last_run[0]
    st = run["table"][0, 0]
    st = st[st >= 0]
    tok = s.get_init_msa(msa, L, 1).numpy().astype(np.int32)
    tok[0, R - 1, st] = 32
    full = msa_model.model.forward_logits(tok)
    d = np.abs(full[0, 4, st] - run["sampled_logits"][0][:len(st)]).max()
    print("\nconfig 5: pruned Gibbs logits vs unpruned forward: max|diff| = %.3e (logit std %.2f)" % (d, full.std()))
    assert d < 2e-2          # same arithmetic up to the GEMM kernel chosen for 51 rows vs 65 664 rows (bf16 rounding flips)


def test_config5_four_templates_per_call_equal_four_serial_calls(msa_model):
    """Row e5 (BASELINE config 5: 32 templates over 8 GPUs = 4 per GPU): four templates of depth 128 x L = 512 resampled in ONE
    native call per step -- 4 x 65 664 tokens per forward -- give, bit for bit, the logits, tokens and strings of four
    generate_single calls (the reference's loop, pgen_msa_revised.py:107-115), with the same interpreter-RNG consumption and one
    torch seed per template."""
    R, L, steps, passes, burn_in = 128, 512, 10, 3, 2
    s = esm_msa_sampler.ESM_MSA_sampler(msa_model, device="cuda:0")
    s.record = True
    msas = [_template_msa(R, L, seed=100 + i) for i in range(4)]
    excl = [None, [0, 7, 300], None, list(range(500, 512))]
    random.seed(21)
    torch.manual_seed(4)
    serial, runs = [], []
    for m, e in zip(msas, excl):
        serial.append(s.generate_single(list(m), steps=steps, passes=passes, burn_in=burn_in, target_index=0, k=1, exclude_positions=e))
        runs.append(s.last_run[0])
    state = random.getrandbits(32)
    random.seed(21)
    torch.manual_seed(4)
    batched = s.generate_single_batch([list(m) for m in msas], steps=steps, passes=passes, burn_in=burn_in, target_index=0, k=1,
                                      exclude_positions=excl, max_batch=4)
    assert random.getrandbits(32) == state
    assert batched == serial
    for one, many in zip(runs, s.last_run):
        P = one["table"].shape[-1]
        assert (many["table"][:, :, :P] == one["table"]).all()
        valid = one["table"][:, 0] >= 0
        assert np.array_equal(many["sampled_logits"][:, :P][valid], one["sampled_logits"][valid])       # bit for bit
        assert (many["sampled_tokens"][:, :P][valid] == one["sampled_tokens"][valid]).all()
        assert (many["tokens"] == one["tokens"]).all()
    assert len(set(serial)) == 4


_DEFAULT_SHAPE_CHILD = r"""
import random, sys, warnings, numpy as np, torch
sys.path.insert(0, %r)
from protein_gibbs_sampler_amd import esm_msa_sampler, models, weights
cfg = dict(weights.MSA1B_CONFIG)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = models.ESM_MSA1(state_dict=weights.synthetic_state_dict(cfg, seed=2, std=0.025, embed_std=0.3, ln_jitter=0.1), config=cfg)
def template(R, L, seed):
    rng = np.random.default_rng(seed)
    rows = np.asarray(list("ACDEFGHIKLMNPQRSTVWY"))[rng.integers(0, 20, (R, L))]
    rows[rng.random((R, L)) < 0.1] = "-"
    return ["".join(r) for r in rows]
s = esm_msa_sampler.ESM_MSA_sampler(m, device="cuda:0")
s.record = True
msas = [template(32, 299, 300 + i) for i in range(4)]
random.seed(9); torch.manual_seed(3)
out = s.generate_single_batch([list(x) for x in msas], steps=1, passes=2, burn_in=1, target_index=0, k=1, max_batch=int(sys.argv[1]))
np.savez(sys.argv[2], strings=np.asarray(out), logits=np.stack([r["sampled_logits"] for r in s.last_run]),
         tokens=np.stack([r["sampled_tokens"] for r in s.last_run]), final=np.stack([r["tokens"] for r in s.last_run]))
"""


def test_default_pgen_msa_shape_batched_chunked_and_serial_agree(tmp_path):
    """the default pgen_msa_revised shape (alignment_size 32 x ~300 columns: 9.6 k token rows per template) sends the
    N = 768 projections of ONE template to the 64^2 / 128^2 kernels and those of FOUR to the 256^2 + tail-tile launch; identity
    rests on every tile kernel walking k in the same order.  steps = 1 makes every step draw 299 positions (> 256 draws per
    call).  Three child processes: one template per call, four per call, and four per call with the split-R scratch limit at
    1 MB, which forces pg_msa_gibbs_single_batch_run to chunk the batch -- logits, tokens and strings must agree bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, mb, env_extra in (("serial", 1, {}), ("batched", 4, {}), ("chunked", 4, {"PGIBBS_SPLIT_SCRATCH_MB": "1"})):
        out = tmp_path / (tag + ".npz")
        p = subprocess.run([sys.executable, "-c", _DEFAULT_SHAPE_CHILD % root, str(mb), str(out)], capture_output=True, text=True,
                           env=dict(os.environ, **env_extra), timeout=1200)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        res[tag] = np.load(out)
    assert res["serial"]["logits"].shape[1:3] == (2, 299)          # 2 steps (passes) x 299 draws each
    for tag in ("batched", "chunked"):
        assert list(res[tag]["strings"]) == list(res["serial"]["strings"])
        assert np.array_equal(res[tag]["logits"].view(np.uint32), res["serial"]["logits"].view(np.uint32))
        assert (res[tag]["tokens"] == res["serial"]["tokens"]).all() and (res[tag]["final"] == res["serial"]["final"]).all()
    assert len(set(res["serial"]["strings"])) == 4


# ---- the reference's plug-in protocol ------------------------------------------------------------------------------
def test_native_model_call_protocol_esm():
    """`self.model.model(batch)["logits"]` exactly as /root/reference/src/pgen/esm_sampler.py:223 issues it: an int64 torch
    batch (on the sampler's device) in, a dict with a float "logits" tensor [B, T, V] out, after .eval() and .to(device)."""
    cfg = weights.make_config(weights.ESM1B_CONFIG, d_model=256, n_layers=3, d_ffn=512, max_positions=64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = models.ESM1b(state_dict=weights.synthetic_state_dict(cfg, seed=4, std=0.05, embed_std=0.3), config=cfg)
    s = esm_sampler.ESM_sampler(model, device="gpu")
    assert s.model.model.eval() is s.model.model
    batch = s.get_init_seq("MEPAATGQEAEECAHSGRGEAWEEV", 30, 3)                 # int64 [3, 32], <mask>-padded
    assert batch.dtype == torch.int64
    want = model.model.forward_logits(batch.numpy())
    for dev in ("cpu", "cuda:0"):
        out = s.model.model(batch.to(dev))
        assert set(out) == {"logits"}
        lg = out["logits"]
        assert isinstance(lg, torch.Tensor) and lg.dtype == torch.float32 and tuple(lg.shape) == (3, 32, 33)
        assert lg.device.type == torch.device(dev).type
        assert (lg.cpu().numpy() == want).all()
    # what generate_step reads (esm_sampler.py:23): out[gen_idx] for one chain
    assert torch.equal(s.model.model(batch)["logits"][1][5], torch.from_numpy(want[1, 5]))


def test_native_model_call_protocol_msa():
    cfg = weights.make_config(weights.MSA1B_CONFIG, d_model=128, n_layers=2, d_ffn=256, max_positions=64, max_msa_rows=8)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = models.ESM_MSA1(state_dict=weights.synthetic_state_dict(cfg, seed=4, std=0.05, embed_std=0.3), config=cfg)
    s = esm_msa_sampler.ESM_MSA_sampler(model, device="cuda:0")
    batch = s.get_init_msa(["ACDEFGHIKL", "AC-EFGHIKL", "ACDEFGH-KL"], 12, 2)   # int64 [2, 3, 13]
    want = model.model.forward_logits(batch.numpy())
    lg = s.model.model(batch.to("cuda:0"))["logits"]
    assert tuple(lg.shape) == (2, 3, 13, 33) and lg.device.type == "cuda" and (lg.cpu().numpy() == want).all()
