"""Token tables and batch converters of the ESM-1b / ESM-MSA-1b model families.

The reference gets these from fair-esm (`esm.data.Alphabet`, `BatchConverter`) through its model
wrappers (/root/reference/src/pgen/models.py:59-88) and patches the MSA converter so that a literal
"<mask>" in a seed string counts as one column (models.py:6-56).  fair-esm is not a dependency of
this package; the tables are restated from its published vocabulary and pinned by
the reference's own tokenisation tests (test_esm_msa_sampler.py:43-84: <cls>=0, A=5, C=23, D=13,
E=9, B=25, <mask>=32).
"""
import re

import numpy as np
import torch

PROTEINSEQ_TOKS = list("LAGVSERTIDPKQNFYMHWCXBUZO.-")
_TOKEN_RE = re.compile(r"<[a-z_0-9]+>|.")


class Alphabet:
    """"ESM-1b" style alphabet: <cls> <pad> <eos> <unk> + 27 residues/gap symbols + <null_1> + <mask> (33 tokens), or, with
    arch="ESM-1" (esm1_t6 / t12 / t34: pgen.models.ESM6 / ESM12 / ESM34), fair-esm's "ESM-1" table: <null_0> <pad> <eos> <unk> +
    the same 27 symbols + <null_1> + <cls> <mask> <sep> (35 tokens: <cls> = 32, <mask> = 33; pinned by the reference's
    test_esm_sampler.py:46,53: "AA" + 3 masks -> [32, 5, 5, 33, 33, 33])."""

    def __init__(self, prepend_bos=True, append_eos=True, arch="ESM-1b"):
        self.standard_toks = list(PROTEINSEQ_TOKS)
        if arch == "ESM-1":
            self.prepend_toks = ["<null_0>", "<pad>", "<eos>", "<unk>"]
            self.append_toks = ["<cls>", "<mask>", "<sep>"]
        else:
            self.prepend_toks = ["<cls>", "<pad>", "<eos>", "<unk>"]
            self.append_toks = ["<mask>"]
        self.all_toks = list(self.prepend_toks) + list(self.standard_toks)
        while len(self.all_toks) % 8:
            self.all_toks.append("<null_%d>" % (8 - len(self.all_toks) % 8))
        self.all_toks += self.append_toks
        self.tok_to_idx = {t: i for i, t in enumerate(self.all_toks)}
        self.unk_idx = self.tok_to_idx["<unk>"]
        self.padding_idx = self.tok_to_idx["<pad>"]
        self.cls_idx = self.tok_to_idx["<cls>"]
        self.mask_idx = self.tok_to_idx["<mask>"]
        self.eos_idx = self.tok_to_idx["<eos>"]
        self.prepend_bos = prepend_bos
        self.append_eos = append_eos
        # byte -> token id for the one-character tokens: strings without "<" (no literal special tokens) are encoded by one table
        # lookup instead of a regular-expression pass and a dict lookup per character (a 128 x 512 alignment: 200 ms -> 2 ms)
        self._byte_lut = np.full(256, self.unk_idx, dtype=np.int64)
        for t, i in self.tok_to_idx.items():
            if len(t) == 1 and ord(t) < 128:
                self._byte_lut[ord(t)] = i

    def __len__(self):
        return len(self.all_toks)

    def get_idx(self, tok):
        return self.tok_to_idx.get(tok, self.unk_idx)

    def get_tok(self, ind):
        return self.all_toks[int(ind)]

    def to_dict(self):
        return dict(self.tok_to_idx)

    def tokenize(self, text):
        return _TOKEN_RE.findall(text)

    def encode(self, text):
        return [self.get_idx(t) for t in self.tokenize(text)]

    def encode_array(self, text):
        """encode() as an int64 array."""
        if _plain(text):
            return self._byte_lut[np.frombuffer(text.encode("ascii"), dtype=np.uint8)]
        return np.asarray(self.encode(text), dtype=np.int64)

    def decode_rows(self, rows):
        """["".join(get_tok(t) for t in row) for row in rows] for a 2-D integer array or nested list (one table lookup per row)."""
        arr = np.asarray(rows)
        if arr.ndim != 2 or arr.dtype.kind not in "iu":
            return ["".join(self.get_tok(t) for t in row) for row in rows]
        if not hasattr(self, "_tok_table"):
            self._tok_table = np.asarray(self.all_toks, dtype=object)
            self._char_table = np.asarray([ord(t) if len(t) == 1 and ord(t) < 128 else 0 for t in self.all_toks], dtype=np.uint8)
        if arr.size and (arr.min() < 0 or arr.max() >= len(self.all_toks)):
            return ["".join(self.get_tok(t) for t in row) for row in rows]      # raises like the per-token form
        chars = self._char_table[arr]
        if chars.all():                                  # only one-character tokens: the rows are byte strings
            return [r.tobytes().decode("ascii") for r in chars]
        return ["".join(r) for r in self._tok_table[arr]]

    def get_batch_converter(self, msa=False):
        return MSABatchConverter(self) if msa else BatchConverter(self)


def _plain(text):
    """True when every character of `text` is one token for _TOKEN_RE: ASCII, no "<" (a literal special token) and no newline
    (which "." does not match)."""
    return "<" not in text and "\n" not in text and text.isascii()


def rawbatchlen(raw):
    """Number of tokens in a string where <...> counts as one (models.py:6-16)."""
    return len(raw) if _plain(raw) else len(_TOKEN_RE.findall(raw))


class BatchConverter:
    """list[(label, str)] -> (labels, strs, int64 tokens [B, maxlen + bos + eos]) padded with <pad>."""

    def __init__(self, alphabet):
        self.alphabet = alphabet

    def __call__(self, raw_batch):
        a = self.alphabet
        labels, strs = [l for l, _ in raw_batch], [s for _, s in raw_batch]
        enc = [a.encode_array(s) for s in strs]
        max_len = max((len(e) for e in enc), default=0)
        bos = int(a.prepend_bos)
        tokens = np.full((len(raw_batch), max_len + bos + int(a.append_eos)), a.padding_idx, dtype=np.int64)
        for i, e in enumerate(enc):
            if a.prepend_bos:
                tokens[i, 0] = a.cls_idx
            tokens[i, bos:len(e) + bos] = e
            if a.append_eos:
                tokens[i, len(e) + bos] = a.eos_idx
        return labels, strs, torch.from_numpy(tokens)


class MSABatchConverter(BatchConverter):
    """One MSA (list of (label, str)) or a list of MSAs -> int64 [B, R, C]; ragged rows raise
    RuntimeError exactly as in models.py:44-49."""

    def __call__(self, inputs):
        if isinstance(inputs[0][0], str):
            raw_batch = [inputs]
        else:
            raw_batch = inputs
        a = self.alphabet
        batch_size = len(raw_batch)
        max_alignments = max(len(msa) for msa in raw_batch)
        max_seqlen = max(rawbatchlen(msa[0][1]) for msa in raw_batch)
        tokens = np.full((batch_size, max_alignments, max_seqlen + int(a.prepend_bos) + int(a.append_eos)), a.padding_idx,
                         dtype=np.int64)
        labels, strs = [], []
        for i, msa in enumerate(raw_batch):
            if len(set(rawbatchlen(seq) for _, seq in msa)) != 1:
                raise RuntimeError("Received unaligned sequences for input to MSA, all sequence lengths must be equal.")
            msa_labels, msa_strs, msa_tokens = super().__call__(msa)
            labels.append(msa_labels)
            strs.append(msa_strs)
            tokens[i, :msa_tokens.size(0), :msa_tokens.size(1)] = msa_tokens.numpy()
        return labels, strs, torch.from_numpy(tokens)
