#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

Runs ONLY in the build container (needs /root/reference and, for the HF
cross-check, `transformers`); the GPU box and the test-suite only read the
resulting .json/.npz files.  Nothing from the reference is copied: the reference
modules are imported from where they lie and driven with a stand-in model; the
fixtures hold inputs and the outputs the reference produced.

  python tests/golden/make_golden.py            # regenerate everything

Fixtures:
  sampler_esm_*.json   -- pgen.esm_sampler.ESM_sampler.generate driven with a deterministic
                          stand-in model: per-forward token buffers (expose mask scatter and
                          write-back), target indexes per iteration, output strings
  sampler_msa_*.json   -- same for pgen.esm_msa_sampler.ESM_MSA_sampler.generate / generate_single
  misc_ref.json        -- partition(), clean_seed_seq errors, calculate_indexes, in-order indexes
  esm_hf_*.npz         -- HuggingFace EsmForMaskedLM logits for seeded synthetic weights
                          (independent corroboration of oracle/esm_forward.py)
"""
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_SRC = "/root/reference/src"


# --------------------------------------------------------------------------------------
# stand-in model objects for driving the reference
# --------------------------------------------------------------------------------------
def standin_tables():
    """Tables of the stand-in logits function; numpy PCG64 so tests can regenerate them anywhere."""
    g = np.random.default_rng(1234)
    return (g.standard_normal((40, 33), dtype=np.float32), g.standard_normal((2048, 33), dtype=np.float32))


def _standin(arch, context=False):
    """arch: 'esm1b' (bos+eos, mask=32) or 'msa1b' (bos only).  context=True (MSA only): the logits of a row also depend on
    the other rows of its MSA (mean over rows), so a wrong context alignment changes the scores."""
    import re
    import torch

    toks = ["<cls>", "<pad>", "<eos>", "<unk>"] + list("LAGVSERTIDPKQNFYMHWCXBUZO.-") + ["<null_1>", "<mask>"]

    class Alphabet:
        all_toks = toks
        standard_toks = list("LAGVSERTIDPKQNFYMHWCXBUZO.-")
        tok_to_idx = {t: i for i, t in enumerate(toks)}
        padding_idx = 1
        cls_idx = 0
        eos_idx = 2
        mask_idx = 32
        prepend_bos = True
        append_eos = arch == "esm1b"

        def get_idx(self, t):
            return self.tok_to_idx[t]

        def get_tok(self, i):
            return self.all_toks[int(i)]

    alphabet = Alphabet()
    splitter = re.compile(r"<[a-z_0-9]+>|.")

    def encode(s):
        return [alphabet.get_idx(t) for t in splitter.findall(s)]

    def convert_rows(rows):
        enc = [encode(s) for _, s in rows]
        L = max(len(e) for e in enc)
        out = torch.full((len(rows), L + 1 + int(alphabet.append_eos)), alphabet.padding_idx, dtype=torch.int64)
        for i, e in enumerate(enc):
            out[i, 0] = alphabet.cls_idx
            out[i, 1:1 + len(e)] = torch.tensor(e, dtype=torch.int64)
            if alphabet.append_eos:
                out[i, 1 + len(e)] = alphabet.eos_idx
        return out

    def batch_converter(inputs):
        if arch == "esm1b":
            t = convert_rows(inputs)
            return [l for l, _ in inputs], [s for _, s in inputs], t
        raw = [inputs] if isinstance(inputs[0][0], str) else inputs
        # as the reference's patched MSABatchConverter (/root/reference/src/pgen/models.py:23-55): one tensor of the deepest /
        # widest MSA, <pad> elsewhere
        conv = [convert_rows(m) for m in raw]
        out = torch.full((len(conv), max(c.shape[0] for c in conv), max(c.shape[1] for c in conv)), alphabet.padding_idx, dtype=torch.int64)
        for i, c in enumerate(conv):
            out[i, :c.shape[0], :c.shape[1]] = c
        return None, None, out

    class Net(torch.nn.Module):
        """Deterministic logits: a fixed pseudo-random function of (token at position, position,
        left/right neighbour tokens) so outputs depend on the evolving buffer."""

        def __init__(self):
            super().__init__()
            tab, ptab = standin_tables()
            self.tab = torch.from_numpy(tab)
            self.ptab = torch.from_numpy(ptab)
            self.calls = []

        def forward(self, tokens):
            self.calls.append(tokens.detach().cpu().clone())
            L = tokens.shape[-1]
            left = torch.roll(tokens, 1, dims=-1)
            right = torch.roll(tokens, -1, dims=-1)
            logits = self.tab[tokens] + 0.5 * self.tab[left].roll(3, -1) + 0.25 * self.tab[right].roll(7, -1) \
                + self.ptab[:L]
            if context:
                logits = logits + 0.75 * self.tab[tokens].mean(dim=-3, keepdim=True).roll(5, -1)
            return {"logits": logits}

    class Model:
        pass

    m = Model()
    m.model = Net()
    m.alphabet = alphabet
    m.batch_converter = batch_converter
    return m


def _tolist(t):
    return t.tolist()


def gen_sampler_esm():
    sys.path.insert(0, REF_SRC)
    sys.dont_write_bytecode = True
    import torch
    from pgen import esm_sampler as ref

    seed25 = "MEPAATGQEAEECAHSGRGEAWEEV"   # README.md:74 seed, config 1
    rng = np.random.default_rng(1234)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    seed64 = "".join(aa[i] for i in rng.integers(0, 20, 64))
    cases = {
        # config-1 plumbing shape: L=25, 20 iters, 10 % positions (P=2: set path of random.sample), top_k=1 burnin=10
        "cfg1": dict(pyseed=0, n_samples=1, seed_seq=seed25, kw=dict(batch_size=1, num_iters=20, burnin=10, mask=True,
                     in_order=False, num_positions_percent=10, top_k=1)),
        # all draws deterministic (burnin=0, top_k=1): strings comparable end-to-end
        "det_random_pool": dict(pyseed=1, n_samples=5, seed_seq=seed64, kw=dict(batch_size=3, num_iters=6, burnin=0, mask=True,
                                in_order=False, num_positions=16, top_k=1)),
        "det_in_order_leader": dict(pyseed=2, n_samples=4, seed_seq="ACDEFGHIKL", kw=dict(batch_size=2, num_iters=3, burnin=0, mask=True,
                                    in_order=True, num_positions=2, leader_length=3, top_k=1)),
        "det_in_order_rollover": dict(pyseed=2, n_samples=2, seed_seq="ACDEFGHIKL", kw=dict(batch_size=2, num_iters=7, burnin=0, mask=True,
                                      in_order=True, num_positions=3, leader_length=4, rollover_from_start=True, top_k=1)),
        "det_all_positions_nomask": dict(pyseed=3, n_samples=2, seed_seq="ACDEFGHIKLMN", kw=dict(batch_size=2, num_iters=2, burnin=0,
                                         mask=False, num_positions=0, top_k=1)),
        "det_indexes_given": dict(pyseed=4, n_samples=3, seed_seq="ACDEFGHIKLMNPQRS", kw=dict(batch_size=3, num_iters=4, burnin=0, mask=True,
                                  indexes=[2, 5, 7, 11, 13], num_positions=2, top_k=1)),
        "det_maxlen_pad": dict(pyseed=5, n_samples=2, seed_seq="ACD", kw=dict(batch_size=2, max_len=9, num_iters=3, burnin=0, mask=True,
                               num_positions=3, top_k=1)),
        "det_list_seed": dict(pyseed=6, n_samples=5, seed_seq=["ACDEF", "GHI", "KLMNPQR"], kw=dict(batch_size=2, num_iters=2, burnin=0,
                              mask=True, num_positions=2, top_k=1)),
        "det_leader_pct_neg": dict(pyseed=7, n_samples=2, seed_seq="ACDEFGHIKLMNPQRSTVWY", kw=dict(batch_size=2, num_iters=2, burnin=0,
                                   mask=True, num_positions_percent=25, leader_length_percent=20, top_k=1)),
        "det_empty_seed": dict(pyseed=8, n_samples=4, seed_seq="", kw=dict(batch_size=4, max_len=10, num_iters=3, burnin=0, top_k=1)),
        "det_temperature": dict(pyseed=9, n_samples=2, seed_seq="ACDEFGHIKL", kw=dict(batch_size=2, num_iters=2, burnin=0, mask=True,
                                num_positions=4, top_k=1, temperature=0.7)),
        "det_numpos_clamp": dict(pyseed=10, n_samples=3, seed_seq="ACDEFG", kw=dict(batch_size=2, num_iters=2, burnin=0, mask=True,
                                 num_positions=50, leader_length=2, top_k=1)),
    }
    out = {}
    for name, c in cases.items():
        model = _standin("esm1b")
        s = ref.ESM_sampler(model, device="cpu")
        rec_targets = []
        orig_rand, orig_order = s.get_random_target_index, s.get_target_index_in_order

        def wrap_rand(*a, _o=orig_rand, **k):
            r = _o(*a, **k)
            rec_targets.append([list(x) for x in r])
            return r

        def wrap_order(*a, _o=orig_order, **k):
            li, r = _o(*a, **k)
            rec_targets.append([list(x) for x in r])
            return li, r

        s.get_random_target_index = wrap_rand
        s.get_target_index_in_order = wrap_order
        random.seed(c["pyseed"])
        torch.manual_seed(c["pyseed"])
        kw = dict(c["kw"])
        strings = s.generate(c["n_samples"], c["seed_seq"], show_progress_bar=False, **kw)
        out[name] = dict(pyseed=c["pyseed"], n_samples=c["n_samples"], seed_seq=c["seed_seq"], kw=c["kw"],
                         forward_inputs=[_tolist(t) for t in model.model.calls],
                         targets=rec_targets, strings=strings,
                         py_state_after=list(random.getstate()[1][-1:]) + [random.getrandbits(32)])
    with open(os.path.join(HERE, "sampler_esm.json"), "w") as f:
        json.dump(out, f)
    print("sampler_esm.json:", {k: len(v["forward_inputs"]) for k, v in out.items()})


def gen_sampler_msa():
    sys.path.insert(0, REF_SRC)
    import torch
    from pgen import esm_msa_sampler as ref

    msa4 = ["ACDEFGHIKL", "AC-EFGHIKL", "ACDEFG--KL", "MCDEFGHIKV"]
    cases = {
        "det_random": dict(pyseed=11, n_samples=8, seed_msa=msa4, kw=dict(batch_size=2, num_iters=3, burnin=0, mask=True,
                           num_positions=3, top_k=1)),
        "det_in_order": dict(pyseed=12, n_samples=4, seed_msa=msa4, kw=dict(batch_size=1, num_iters=4, burnin=0, mask=True,
                             in_order=True, num_positions=2, leader_length=2, top_k=1)),
        "det_all_positions": dict(pyseed=13, n_samples=4, seed_msa=msa4[:2], kw=dict(batch_size=2, num_iters=2, burnin=0, mask=False,
                                  num_positions=0, top_k=1)),
        "det_maxlen_pct": dict(pyseed=14, n_samples=6, seed_msa=["AAA", "AAC"], kw=dict(batch_size=2, max_len=5, num_iters=2, burnin=0,
                               mask=True, num_positions_percent=70, top_k=1)),
        "det_two_rounds": dict(pyseed=15, n_samples=10, seed_msa=msa4[:3], kw=dict(batch_size=2, num_iters=2, burnin=0, mask=True,
                               num_positions=2, leader_length=1, in_order=True, top_k=1)),
    }
    out = {}
    for name, c in cases.items():
        model = _standin("msa1b")
        s = ref.ESM_MSA_sampler(model, device="cpu")
        rec = []
        o_r, o_o, o_a = s.get_random_target_index, s.get_target_index_in_order, s.get_target_indexes_all_positions

        def w_r(*a, _o=o_r, **k):
            r = _o(*a, **k)
            rec.append([[list(x) for x in b] for b in r])
            return r

        def w_o(*a, _o=o_o, **k):
            li, r = _o(*a, **k)
            rec.append([[list(x) for x in b] for b in r])
            return li, r

        def w_a(*a, _o=o_a, **k):
            r = _o(*a, **k)
            rec.append([[list(x) for x in b] for b in r])
            return r

        s.get_random_target_index, s.get_target_index_in_order, s.get_target_indexes_all_positions = w_r, w_o, w_a
        random.seed(c["pyseed"])
        torch.manual_seed(c["pyseed"])
        strings = s.generate(c["n_samples"], c["seed_msa"], show_progress_bar=False, **c["kw"])
        out[name] = dict(pyseed=c["pyseed"], n_samples=c["n_samples"], seed_msa=c["seed_msa"], kw=c["kw"],
                         forward_inputs=[_tolist(t) for t in model.model.calls], targets=rec, strings=strings,
                         py_state_after=[random.getrandbits(32)])
    # generate_single (esm_msa_sampler.py:101-147), incl. quirk Q2 (row -1 masked) and legacy target_index=-1
    single = {
        "single_default_det": dict(pyseed=21, seed_msa=msa4, kw=dict(steps=3, passes=2, burn_in=0, target_index=0, k=1)),
        "single_legacy_last": dict(pyseed=22, seed_msa=msa4, kw=dict(steps=4, passes=2, burn_in=0, target_index=-1, k=1)),
        "single_exclude": dict(pyseed=23, seed_msa=msa4, kw=dict(steps=2, passes=3, burn_in=0, target_index=0, k=1,
                               exclude_positions=[0, 3, 4])),
        "single_steps_gt_len": dict(pyseed=24, seed_msa=["ACD", "ACE", "GCD"], kw=dict(steps=10, passes=2, burn_in=0, target_index=1, k=1)),
    }
    for name, c in single.items():
        model = _standin("msa1b")
        s = ref.ESM_MSA_sampler(model, device="cpu")
        random.seed(c["pyseed"])
        torch.manual_seed(c["pyseed"])
        string = s.generate_single(list(c["seed_msa"]), **c["kw"])
        out[name] = dict(pyseed=c["pyseed"], seed_msa=c["seed_msa"], kw=c["kw"],
                         forward_inputs=[_tolist(t) for t in model.model.calls], string=string,
                         py_state_after=[random.getrandbits(32)])
    with open(os.path.join(HERE, "sampler_msa.json"), "w") as f:
        json.dump(out, f)
    print("sampler_msa.json:", {k: len(v["forward_inputs"]) for k, v in out.items()})


def gen_misc():
    sys.path.insert(0, REF_SRC)
    from pgen import esm_sampler as ref
    from pgen import esm_msa_sampler as refm
    out = {}
    out["partition"] = [dict(n=n, parts=p, out=refm.partition(list(range(1, n + 1)), p))
                        for n in (1, 7, 10, 51, 512) for p in (1, 2, 3, 4, 7, 10, 11, 600)]
    try:  # empty input: the reference divides by zero (esm_msa_sampler.py:19)
        refm.partition([], 3)
        out["partition_empty"] = "ok"
    except ZeroDivisionError:
        out["partition_empty"] = "ZeroDivisionError"
    errs = {}
    for s in ("X", "AXB", "A.C", "A-C", "ac*"):
        try:
            ref.ESM_sampler.clean_seed_seq(s)
            errs[s] = None
        except Exception as e:  # noqa
            errs[s] = sorted(str(e)[len("Invalid input character: "):].split(","))
    out["esm_clean_errors"] = errs
    # python RNG stream samples: (seed, n, k) -> random.sample(range(1, n+1), k) x 3 ; shuffle ; choices
    streams = []
    for sd in (0, 1, 12345, 2**40 + 17):
        for (n, k) in ((25, 2), (256, 25), (257, 25), (512, 51), (10, 10), (21, 6), (22, 6), (85, 6), (86, 6), (5, 0)):
            random.seed(sd)
            a = [random.sample(range(1, n + 1), k) for _ in range(3)]
            lst = list(range(n))
            random.shuffle(lst)
            ch = random.choices(["a", "b", "c"], k=5)
            streams.append(dict(seed=sd, n=n, k=k, samples=a, shuffled=lst, choices=ch, next32=random.getrandbits(32)))
    out["py_streams"] = streams
    with open(os.path.join(HERE, "misc_ref.json"), "w") as f:
        json.dump(out, f)
    print("misc_ref.json written")


# --------------------------------------------------------------------------------------
# HuggingFace cross-check of the ESM-1b forward restatement
# --------------------------------------------------------------------------------------
HF_CASES = {
    # name: (cfg kwargs, weight seed, std, embed_std, ln_jitter, token rows)
    "tiny": (dict(d_model=128, n_layers=2, n_heads=2, d_ffn=256, max_pos=64), 7, 0.08, 0.5, 0.1),
    "small": (dict(d_model=256, n_layers=4, n_heads=4, d_ffn=512, max_pos=300), 11, 0.05, 0.3, 0.1),
    # the REAL ESM-1b sizes (33 layers, d = 1280, 20 heads, FFN 5120, 1024 positions; 650 M parameters) at a realistic logit
    # scale: pins the full-size oracle -- and through it the engine -- to the independent implementation, not only small models
    "full": (dict(d_model=1280, n_layers=33, n_heads=20, d_ffn=5120, max_pos=1024), 11, 0.025, 0.3, 0.1),
}


def hf_tokens(name):
    rng = np.random.default_rng(99)
    if name == "tiny":
        B, L = 3, 25
    elif name == "full":
        B, L = 2, 256          # config 2's chain shape (T = 258)
    else:
        B, L = 2, 256
    tok = rng.integers(4, 24, size=(B, L + 2))
    tok[:, 0] = 0
    tok[:, -1] = 2
    for b in range(B):
        pos = rng.choice(np.arange(1, L + 1), size=max(1, (b + 1) * L // 10), replace=False)
        tok[b, pos] = 32
    return tok


def gen_hf(only=None):
    import torch
    from transformers import EsmConfig as HC, EsmForMaskedLM
    from oracle.esm_forward import EsmConfig, synthetic_esm_weights, esm1b_forward

    for name, (ck, seed, std, estd, jit) in HF_CASES.items():
        if only and name not in only:
            continue
        cfg = EsmConfig(**ck)
        w = synthetic_esm_weights(cfg, seed=seed, std=std, embed_std=estd, ln_jitter=jit)
        hc = HC(vocab_size=33, hidden_size=cfg.d_model, num_hidden_layers=cfg.n_layers, num_attention_heads=cfg.n_heads,
                intermediate_size=cfg.d_ffn, max_position_embeddings=cfg.max_pos + 2, position_embedding_type="absolute",
                emb_layer_norm_before=True, token_dropout=True, mask_token_id=32, pad_token_id=1, layer_norm_eps=1e-5,
                hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        m = EsmForMaskedLM(hc).eval()
        sd = {}
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        sd["esm.embeddings.word_embeddings.weight"] = T(w["embed_tokens.weight"])
        sd["esm.embeddings.position_embeddings.weight"] = T(w["embed_positions.weight"])
        sd["esm.embeddings.layer_norm.weight"] = T(w["emb_layer_norm_before.weight"])
        sd["esm.embeddings.layer_norm.bias"] = T(w["emb_layer_norm_before.bias"])
        for i in range(cfg.n_layers):
            p, q = "layers.%d." % i, "esm.encoder.layer.%d." % i
            for a, b in (("q_proj", "attention.self.query"), ("k_proj", "attention.self.key"), ("v_proj", "attention.self.value"),
                         ("out_proj", "attention.output.dense")):
                sd[q + b + ".weight"] = T(w[p + "self_attn." + a + ".weight"])
                sd[q + b + ".bias"] = T(w[p + "self_attn." + a + ".bias"])
            sd[q + "attention.LayerNorm.weight"] = T(w[p + "self_attn_layer_norm.weight"])
            sd[q + "attention.LayerNorm.bias"] = T(w[p + "self_attn_layer_norm.bias"])
            sd[q + "intermediate.dense.weight"] = T(w[p + "fc1.weight"])
            sd[q + "intermediate.dense.bias"] = T(w[p + "fc1.bias"])
            sd[q + "output.dense.weight"] = T(w[p + "fc2.weight"])
            sd[q + "output.dense.bias"] = T(w[p + "fc2.bias"])
            sd[q + "LayerNorm.weight"] = T(w[p + "final_layer_norm.weight"])
            sd[q + "LayerNorm.bias"] = T(w[p + "final_layer_norm.bias"])
        sd["esm.encoder.emb_layer_norm_after.weight"] = T(w["emb_layer_norm_after.weight"])
        sd["esm.encoder.emb_layer_norm_after.bias"] = T(w["emb_layer_norm_after.bias"])
        sd["lm_head.dense.weight"] = T(w["lm_head.dense.weight"])
        sd["lm_head.dense.bias"] = T(w["lm_head.dense.bias"])
        sd["lm_head.layer_norm.weight"] = T(w["lm_head.layer_norm.weight"])
        sd["lm_head.layer_norm.bias"] = T(w["lm_head.layer_norm.bias"])
        sd["lm_head.bias"] = T(w["lm_head.bias"])
        sd["lm_head.decoder.weight"] = T(w["embed_tokens.weight"])
        missing, unexpected = m.load_state_dict(sd, strict=False)
        missing = [k for k in missing if "contact_head" not in k and "position_ids" not in k and "inv_freq" not in k]
        assert not missing and not unexpected, (missing, unexpected)
        tok = hf_tokens(name)
        with torch.no_grad():
            hf_logits = m(input_ids=torch.from_numpy(tok), attention_mask=None).logits.numpy()
        mine = esm1b_forward(w, cfg, tok)
        err = float(np.abs(mine - hf_logits).max())
        print("HF cross-check %s: max|oracle - HF| = %.3e  (logit std %.3f)" % (name, err, hf_logits.std()))
        assert err < 2e-4
        np.savez_compressed(os.path.join(HERE, "esm_hf_%s.npz" % name), tokens=tok.astype(np.int32), logits=hf_logits.astype(np.float32),
                            cfg=json.dumps(ck), seed=seed, std=std, embed_std=estd, ln_jitter=jit)


def gen_loglik():
    """log_likelihood_batch of both reference samplers driven with the stand-in model (next-tier path)."""
    sys.path.insert(0, REF_SRC)
    import torch
    from pgen import esm_msa_sampler as refm
    from pgen import esm_sampler as ref
    out = {"esm": [], "msa": []}
    seqs = ["MRHGDISSSNDTVGVAVVNYKMPRLHTAAEVLDNAR", "ACDEFGHIKL", "MKV"]
    for kw in (dict(with_masking=True), dict(with_masking=True, mask_distance=5), dict(with_masking=True, mask_distance=2, batch_size=1),
               dict(with_masking=False), dict(with_masking=True, mask_distance=50, batch_size=3)):
        s = ref.ESM_sampler(_standin("esm1b"), device="cpu")
        res = [(m, l) for m, l in s.log_likelihood_batch(list(seqs), **kw)]
        out["esm"].append(dict(seqs=seqs, kw={k: (v if v != float("inf") else None) for k, v in kw.items()},
                               means=[r[0] for r in res], lists=[r[1] for r in res]))
    msas = [["ACDEFGHIKL", "AC-EFGHIKL", "ACDEFG--KL", "MCDEFGHIKV"], ["MKV-A", "MKVAA", "M-VAA"]]
    for kw in (dict(target_index=0), dict(target_index=1, mask_distance=3), dict(target_index=2, count_gaps=True, mask_distance=4),
               dict(target_index=0, with_masking=False), dict(target_index=1, with_masking=False, count_gaps=True),
               dict(target_index=-1, mask_distance=2, batch_size=2)):
        # the whole ragged list in ONE call: the unmasked path pads it to one [n, R_max, C_max] tensor (esm_msa_sampler.py:341,
        # 416-431), so the shallower / narrower MSA is scored with <pad> around it -- reproduced by this package since round 4
        s = refm.ESM_MSA_sampler(_standin("msa1b"), device="cpu")
        means, lists, per_msa = [], [], False
        try:
            for m, l in s.log_likelihood_batch([list(msa) for msa in msas], **kw):
                means.append(m)
                lists.append(l)
        except AssertionError:
            # target_index = -1 on a list of different depths: the reference reads tokens[:, -1] of the PADDED tensor (:370), counts
            # the shallower MSA's <pad> row and trips its own `len(counted_idx) == msa_denominator` check -- recorded per MSA instead
            means, lists, per_msa = [], [], True
            for msa in msas:
                s = refm.ESM_MSA_sampler(_standin("msa1b"), device="cpu")
                for m, l in s.log_likelihood_batch([list(msa)], **kw):
                    means.append(m)
                    lists.append(l)
        out["msa"].append(dict(msas=msas, kw=kw, means=means, lists=lists, per_msa=per_msa))
    with open(os.path.join(HERE, "loglik.json"), "w") as f:
        json.dump(out, f)
    print("loglik.json written:", len(out["esm"]), len(out["msa"]))


if __name__ == "__main__":
    which = sys.argv[1:] or ["hf", "misc", "esm", "msa", "loglik"]
    if "hf" in which:
        gen_hf()
    if "hf_full" in which:
        gen_hf(only=("full",))
    if "misc" in which:
        gen_misc()
    if "esm" in which:
        gen_sampler_esm()
    if "msa" in which:
        gen_sampler_msa()
    if "loglik" in which:
        gen_loglik()
