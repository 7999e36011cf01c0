"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the Gibbs control loops of the reference:

  ESM_sampler.generate            /root/reference/src/pgen/esm_sampler.py:128-240
    get_init_seq / clean_seed_seq   :95-126      calculate_indexes  :264-274
    get_random_target_index         :242-246     get_target_index_in_order :248-257
    mask_target_indexes             :259-262     untokenize_batch   :84-93
  ESM_MSA_sampler.generate        /root/reference/src/pgen/esm_msa_sampler.py:151-253
    get_init_msa :78-90, index helpers :266-304, mask_target_indexes :255-259
  ESM_MSA_sampler.generate_single esm_msa_sampler.py:101-147, partition :13-31,
    mask_target_indexes_single :261-264
  MSABatchConverter / rawbatchlen /root/reference/src/pgen/models.py:6-56

including the behavioural quirks of SURVEY.md Appendix C (Q1 `indexes`/`num_positions`
rebinding across batches, Q2 generate_single masks row -1, Q4, Q5, Q6, Q9).

Position selection uses oracle.pyrandom (CPython-exact MT19937); the token draw is
oracle.draw (pg_draw v1).  `forward` is any callable tokens[int ndarray] -> logits fp32.

Pinned by tests/test_oracle_sampler.py against tests/golden/sampler_*.json, which were
produced by importing and running the reference itself (tests/golden/make_golden.py).
"""
import math
import re

import numpy as np

from . import draw as _draw
from .pyrandom import PyRandom

ESM_ALLOWED = "ACDEFGHIKLMNPQRSTVWY"
MSA_ALLOWED = "-ACDEFGHIKLMNPQRSTVWY"

_TOKS_1B = ["<cls>", "<pad>", "<eos>", "<unk>"] + list("LAGVSERTIDPKQNFYMHWCXBUZO.-") + ["<null_1>", "<mask>"]
_SPLIT = re.compile(r"<[a-z_0-9]+>|.")


class Alphabet1b:
    """ESM-1b / MSA-1b token table (pinned by reference tests
    test_esm_msa_sampler.py:45-66: <cls>=0, A=5, C=23, D=13, E=9, B=25, <mask>=32)."""

    def __init__(self, append_eos):
        self.all_toks = list(_TOKS_1B)
        self.tok_to_idx = {t: i for i, t in enumerate(self.all_toks)}
        self.padding_idx, self.cls_idx, self.eos_idx, self.mask_idx = 1, 0, 2, 32
        self.prepend_bos, self.append_eos = True, append_eos

    def get_idx(self, t):
        return self.tok_to_idx[t]

    def get_tok(self, i):
        return self.all_toks[int(i)]

    def encode(self, s):
        return [self.tok_to_idx[t] for t in _SPLIT.findall(s)]

    def rows_to_tokens(self, strs):
        enc = [self.encode(s) for s in strs]
        L = max(len(e) for e in enc)
        out = np.full((len(enc), L + 1 + int(self.append_eos)), self.padding_idx, dtype=np.int64)
        for i, e in enumerate(enc):
            out[i, 0] = self.cls_idx
            out[i, 1:1 + len(e)] = e
            if self.append_eos:
                out[i, 1 + len(e)] = self.eos_idx
        return out


def clean_seed_seq(seq, allowed):
    s = seq.upper()
    bad = set(s) - set(allowed)
    if bad:
        raise Exception("Invalid input character: " + ",".join(bad))
    return s


def partition(input_list, num_partitions):
    """esm_msa_sampler.py:13-31 (empty input divides by zero there too)."""
    if len(input_list) < num_partitions:
        num_partitions = len(input_list)
    q, rem = len(input_list) // num_partitions, len(input_list) % num_partitions
    out, pos = [], 0
    for i in range(num_partitions):
        n = q + (1 if i < rem else 0)
        out.append(list(input_list[pos:pos + n]))
        pos += n
    return out


def calculate_indexes(indexes, leader_length, max_len, rollover_from_start):
    if indexes is None:
        indexes = list(range(1, max_len + 1))
        if not rollover_from_start:
            indexes = indexes[leader_length:]
            last_i = leader_length - 1
        else:
            last_i = -1
    else:
        last_i = -1
    return indexes, last_i


def in_order_targets(indexes, next_i, num_positions):
    t = []
    for _ in range(num_positions):
        next_i = (next_i + 1) % len(indexes)
        t.append(indexes[next_i])
    return next_i, t


class OracleESMSampler:
    def __init__(self, forward, rng=None, draw_seed=0):
        self.forward = forward
        self.alphabet = Alphabet1b(append_eos=True)
        self.rng = rng if rng is not None else PyRandom(0)
        self.valid_aa_idx = sorted(self.alphabet.get_idx(t) for t in ESM_ALLOWED)
        self.draw_seed = draw_seed
        self.trace = {"forward_inputs": [], "targets": [], "logits_rows": []}

    def get_init_seq(self, seed_seq, max_len, batch_size=1):
        if isinstance(seed_seq, list):
            batch = self.rng.choices(seed_seq, k=batch_size)
            strs = [clean_seed_seq(s, ESM_ALLOWED) + "<mask>" * (max_len - len(s)) for s in batch]
        elif isinstance(seed_seq, str):
            s = clean_seed_seq(seed_seq, ESM_ALLOWED)
            strs = [s + "<mask>" * (max_len - len(seed_seq))] * batch_size
        else:
            raise Exception("seed sequence should either be a string or list")
        return self.alphabet.rows_to_tokens(strs)

    def untokenize_batch(self, batch):
        return ["".join(self.alphabet.get_tok(t) for t in row[1:-1]) for row in batch]

    def generate(self, n_samples, seed_seq, batch_size=1, in_order=False, max_len=None, leader_length=0,
                 leader_length_percent=None, top_k=0, temperature=None, num_iters=10, burnin=float("inf"), mask=True,
                 num_positions=0, num_positions_percent=None, indexes=None, rollover_from_start=False, stream0=0):
        if isinstance(seed_seq, str):
            sequence_length = len(seed_seq)
        elif isinstance(seed_seq, list):
            sequence_length = max(len(s) for s in seed_seq)
        else:
            raise ValueError("Unknown seed sequence format, expecting str or list")
        sequences = []
        n_batches = math.ceil(n_samples / batch_size)
        if max_len is None:
            max_len = sequence_length
        if num_positions_percent is not None:
            num_positions = int(max_len * (num_positions_percent / 100))
        num_positions = max(num_positions, 0)
        if leader_length_percent is not None:
            leader_length = int(max_len * (leader_length_percent / 100))
        leader_length = max(leader_length, 0)
        for batch_n in range(n_batches):
            batch = self.get_init_seq(seed_seq, max_len, batch_size)
            indexes, last_i = calculate_indexes(indexes, leader_length, max_len, rollover_from_start)   # Q1: rebinding
            if num_positions > len(indexes):
                num_positions = len(indexes)
            for ii in range(num_iters):
                if num_positions > 0:
                    if in_order:
                        last_i, t = in_order_targets(indexes, last_i, num_positions)
                        targets = [t] * batch_size
                    else:
                        targets = [self.rng.sample(indexes, num_positions) for _ in range(batch_size)]
                    self.trace["targets"].append([list(t) for t in targets])
                else:
                    targets = [list(indexes)] * batch_size
                if mask:
                    for b in range(batch_size):
                        for kk in targets[b]:
                            batch[b][kk] = self.alphabet.mask_idx
                self.trace["forward_inputs"].append(batch.copy())
                out = np.asarray(self.forward(batch), dtype=np.float32)
                rows, rid, slot = [], [], []
                for b in range(batch_size):
                    for p, kk in enumerate(targets[b]):
                        rows.append(out[b, kk])
                        rid.append(batch_n * batch_size + b)
                        slot.append(p)
                if rows:
                    rows = np.stack(rows)
                    self.trace["logits_rows"].append(rows)
                    toks = _draw.draw_rows(rows, self.valid_aa_idx, top_k, ii < burnin, temperature, rid, ii, slot,
                                           stream0, self.draw_seed)
                    i = 0
                    for b in range(batch_size):
                        for kk in targets[b]:
                            batch[b][kk] = toks[i]      # sequential semantics: later duplicates win
                            i += 1
            strs = self.untokenize_batch(batch)
            if batch_n == n_batches - 1:
                sequences += strs[0:n_samples - len(sequences)]
            else:
                sequences += strs
        return sequences


class OracleMSASampler:
    def __init__(self, forward, rng=None, draw_seed=0):
        self.forward = forward
        self.alphabet = Alphabet1b(append_eos=False)
        self.rng = rng if rng is not None else PyRandom(0)
        self.valid_aa_idx = sorted(self.alphabet.get_idx(t) for t in MSA_ALLOWED)
        self.draw_seed = draw_seed
        self.trace = {"forward_inputs": [], "targets": []}

    def get_init_msa(self, seed_msa, max_len, batch_size=1):
        strs = []
        for seq in seed_msa:
            s = clean_seed_seq(seq, MSA_ALLOWED)
            strs.append(s + "<mask>" * (max_len - len(s)))
        lens = {len(_SPLIT.findall(s)) for s in strs}
        if len(lens) != 1:
            raise RuntimeError("Received unaligned sequences for input to MSA, all sequence lengths must be equal.")
        one = self.alphabet.rows_to_tokens(strs)
        return np.stack([one] * batch_size)

    def untokenize_batch(self, batch):
        return ["".join(self.alphabet.get_tok(t) for t in seq[1:]) for msa in batch for seq in msa]

    def generate(self, n_samples, seed_msa, batch_size=1, in_order=False, max_len=None, leader_length=0,
                 leader_length_percent=None, top_k=0, temperature=None, num_iters=10, burnin=float("inf"), mask=True,
                 num_positions=0, num_positions_percent=None, indexes=None, rollover_from_start=False, stream0=0):
        R = len(seed_msa)
        sequence_length = len(seed_msa[0])
        sequences = []
        n_rounds = math.ceil(n_samples / R / batch_size)
        if num_positions_percent is not None:
            num_positions = int(sequence_length * (num_positions_percent / 100))
        num_positions = max(num_positions, 0)
        if leader_length_percent is not None:
            leader_length = int(sequence_length * (leader_length_percent / 100))
        leader_length = max(leader_length, 0)
        if max_len is None:
            max_len = sequence_length
        for rnd in range(n_rounds):
            batch = self.get_init_msa(seed_msa, max_len, batch_size)
            indexes, last_i = calculate_indexes(indexes, leader_length, max_len, rollover_from_start)
            if num_positions > len(indexes):
                num_positions = len(indexes)
            for ii in range(num_iters):
                if num_positions > 0:
                    if in_order:
                        last_i, t = in_order_targets(indexes, last_i, num_positions)
                        targets = [[t] * R for _ in range(batch_size)]
                    else:
                        targets = [[self.rng.sample(indexes, num_positions) for _ in range(R)] for _ in range(batch_size)]
                else:
                    targets = [[list(indexes)] * R for _ in range(batch_size)]
                self.trace["targets"].append([[list(x) for x in b] for b in targets])
                if mask:
                    for b in range(batch_size):
                        for s in range(R):
                            for kk in targets[b][s]:
                                batch[b][s][kk] = self.alphabet.mask_idx
                self.trace["forward_inputs"].append(batch.copy())
                out = np.asarray(self.forward(batch), dtype=np.float32)
                rows, rid, slot = [], [], []
                for b in range(batch_size):
                    for s in range(R):
                        for p, kk in enumerate(targets[b][s]):
                            rows.append(out[b, s, kk])
                            rid.append((rnd * batch_size + b) * R + s)
                            slot.append(p)
                if rows:
                    toks = _draw.draw_rows(np.stack(rows), self.valid_aa_idx, top_k, ii < burnin, temperature, rid, ii,
                                           slot, stream0, self.draw_seed)
                    i = 0
                    for b in range(batch_size):
                        for s in range(R):
                            for kk in targets[b][s]:
                                batch[b][s][kk] = toks[i]
                                i += 1
            strs = self.untokenize_batch(batch)
            if rnd == n_rounds - 1:
                sequences += strs[0:n_samples - len(sequences)]
            else:
                sequences += strs
        return sequences

    def generate_single(self, seed_msa, steps=10, passes=3, burn_in=1, target_index=0, k=1, exclude_positions=None,
                        stream0=0):
        excl = set(i + 1 for i in (exclude_positions or []))
        L = len(seed_msa[0])
        positions = [x for x in range(1, L + 1) if x not in excl]
        batch = self.get_init_msa(seed_msa, L, 1)
        fwd = 0
        for pass_num in range(passes):
            self.rng.shuffle(positions)
            step_indices = partition(positions, steps)
            for step in step_indices:
                for kk in step:
                    batch[0][-1][kk] = self.alphabet.mask_idx            # Q2: row -1, not target_index
                self.trace["forward_inputs"].append(batch.copy())
                out = np.asarray(self.forward(batch), dtype=np.float32)
                rows = np.stack([out[0, target_index, kk] for kk in step])
                tr = target_index % len(seed_msa)
                toks = _draw.draw_rows(rows, self.valid_aa_idx, k, pass_num < burn_in, None, [tr] * len(step), fwd,
                                       list(range(len(step))), stream0, self.draw_seed)
                for kk, t in zip(step, toks):
                    batch[0][target_index][kk] = t
                fwd += 1
        return self.untokenize_batch(batch)[target_index]


# ----------------------------------------------------------------------------------------------------
# masked log-likelihood (next-tier path): /root/reference/src/pgen/esm_sampler.py:288-363,
# /root/reference/src/pgen/esm_msa_sampler.py:319-432.  Pinned by tests/golden/loglik.json (reference run).
# ----------------------------------------------------------------------------------------------------
def _log_softmax(a):
    a = np.asarray(a, dtype=np.float32)
    m = a.max(axis=-1, keepdims=True)
    return (a - m - np.log(np.exp(a - m, dtype=np.float32).sum(axis=-1, keepdims=True, dtype=np.float32))).astype(np.float32)


def esm_log_likelihood_batch(forward, seq_list, with_masking=True, mask_distance=float("inf"), batch_size=None):
    alphabet = Alphabet1b(append_eos=True)
    if batch_size is None:
        batch_size = len(seq_list)
    out = []
    for seq in seq_list:
        s = clean_seed_seq(seq, ESM_ALLOWED)
        one = alphabet.rows_to_tokens([s])
        start, end = 1, len(seq) + 1
        if with_masking:
            n = int(min(mask_distance, len(s)))
            copies = np.repeat(one, n, axis=0)
            for i in range(n):
                copies[i, list(range(start + i, end, n))] = alphabet.mask_idx
        else:
            n, copies = 1, one
        total, lst = np.float32(0.0), []
        for b0 in range(0, n, max(1, batch_size)):
            lp = _log_softmax(forward(copies[b0:b0 + max(1, batch_size)]))
            for i in range(lp.shape[0]):
                for pos in range(start, end):
                    if not with_masking or (pos - start) % n == i + b0:
                        v = lp[i, pos, one[0, pos]]
                        total = np.float32(total + v)
                        lst.append(float(v))
        out.append((float(total / np.float32(len(seq))), lst))
    return out


def msa_log_likelihood_batch(forward, msa_list, target_index=0, with_masking=True, count_gaps=False,
                             mask_distance=float("inf"), batch_size=1):
    alphabet = Alphabet1b(append_eos=False)
    gap = {alphabet.get_idx("-")}
    out = []
    if not with_masking:
        # esm_msa_sampler.py:341, 416-431: the WHOLE list is converted into one tensor padded with <pad> to the deepest / widest
        # MSA, `batch_size` of them per forward; scores are read from the padded tensor's row `target_index`
        rows = [alphabet.rows_to_tokens([clean_seed_seq(s, MSA_ALLOWED) for s in msa]) for msa in msa_list]
        R, C = max(r.shape[0] for r in rows), max(r.shape[1] for r in rows)
        tokens = np.full((len(rows), R, C), alphabet.padding_idx, dtype=np.int64)
        for i, r in enumerate(rows):
            tokens[i, :r.shape[0], :r.shape[1]] = r
        for b0 in range(0, len(rows), max(1, batch_size)):
            lp = _log_softmax(forward(tokens[b0:b0 + max(1, batch_size)]))
            for i in range(lp.shape[0]):
                msa = msa_list[b0 + i]
                L = len(msa[target_index])
                denom = L - (0 if count_gaps else msa[target_index].count("-"))
                orig = tokens[b0 + i, target_index]
                total, lst = np.float32(0.0), []
                for pos in range(1, L + 1):
                    if count_gaps or int(orig[pos]) not in gap:
                        v = lp[i, target_index, pos, orig[pos]]
                        total = np.float32(total + v)
                        lst.append(float(v))
                out.append((float(total / np.float32(denom)), lst))
        return out
    for msa in msa_list:
        one = alphabet.rows_to_tokens([clean_seed_seq(s, MSA_ALLOWED) for s in msa])[None]
        L = len(msa[target_index])
        denom = L - (0 if count_gaps else msa[target_index].count("-"))
        start, end = 1, L + 1
        orig = one[0, target_index]
        if with_masking:
            n = int(min(mask_distance, L))
            copies = np.repeat(one, n, axis=0)
            for i in range(n):
                copies[i, target_index, list(range(start + i, end, n))] = alphabet.mask_idx
        else:
            n, copies = 1, one
        total, lst = np.float32(0.0), []
        for b0 in range(0, n, max(1, batch_size)):
            lp = _log_softmax(forward(copies[b0:b0 + max(1, batch_size)]))
            for i in range(lp.shape[0]):
                for pos in range(start, end):
                    if (not with_masking or (pos - start) % n == i + b0) and (count_gaps or int(orig[pos]) not in gap):
                        v = lp[i, target_index, pos, orig[pos]]
                        total = np.float32(total + v)
                        lst.append(float(v))
        out.append((float(total / np.float32(denom)), lst))
    return out
