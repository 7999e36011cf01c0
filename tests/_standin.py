"""Deterministic stand-in logits function shared by the golden generator and the tests.

Same function as tests/golden/make_golden.py::_standin.Net.forward (the one the reference was
driven with when the fixtures were recorded), in numpy and -- for GPU tests -- as a torch module.
"""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def standin_tables():
    g = np.random.default_rng(1234)
    return (g.standard_normal((40, 33), dtype=np.float32), g.standard_normal((2048, 33), dtype=np.float32))


def standin_logits_np(tokens):
    tab, ptab = standin_tables()
    tokens = np.asarray(tokens)
    L = tokens.shape[-1]
    left = np.roll(tokens, 1, axis=-1)
    right = np.roll(tokens, -1, axis=-1)
    half, quarter = np.float32(0.5), np.float32(0.25)
    a = tab[tokens] + half * np.roll(tab[left], 3, axis=-1)
    a = a + quarter * np.roll(tab[right], 7, axis=-1)
    return (a + ptab[:L]).astype(np.float32)


def make_standin_torch_module(context=False):
    import torch

    tab, ptab = standin_tables()

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.register_buffer("tab", torch.from_numpy(tab))
            self.register_buffer("ptab", torch.from_numpy(ptab))
            self.calls = []

        def forward(self, tokens):
            self.calls.append(tokens.detach().cpu().clone())
            L = tokens.shape[-1]
            left = torch.roll(tokens, 1, dims=-1)
            right = torch.roll(tokens, -1, dims=-1)
            logits = self.tab[tokens] + 0.5 * self.tab[left].roll(3, -1) + 0.25 * self.tab[right].roll(7, -1) \
                + self.ptab[:L]
            if context:                      # MSA fixtures of the caller modules: rows see the column means of their MSA
                logits = logits + 0.75 * self.tab[tokens].mean(dim=-3, keepdim=True).roll(5, -1)
            return {"logits": logits}

    return Net()


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


# --------------------------------------------------------------------------------------
# Deterministic stand-ins for the external programs (phmmer / mafft / muscle are not in the image): the golden
# generator patches them into the reference's pipeline modules, the tests patch the same functions into this
# package's, so the recorded outputs pin everything around them.
# --------------------------------------------------------------------------------------
def fake_run_phmmer(query, database, evalue=10, cpu=2, max_mode=False):
    """Database record names ranked by (shared 2-mers with the query, then name); records sharing none are not hits."""
    names, seqs, cur = [], [], None
    with open(database) as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                names.append(line[1:].split()[0])
                seqs.append("")
            elif line:
                seqs[-1] += line
    q2 = {query[i:i + 2] for i in range(len(query) - 1)}
    scored = []
    for n, s in zip(names, seqs):
        sc = len(q2 & {s[i:i + 2] for i in range(len(s) - 1)})
        if sc > 0:
            scored.append((-sc, n))
    return [n for _, n in sorted(scored)]


def fake_generate_alignment(sequences, ep=0.0, op=1.53):
    """Order-preserving toy aligner: row i is shifted right by (i % 3) gap columns (row 0 by one extra column when there
    are several rows, so the template itself carries gaps) and right-padded with '-' to a common width."""
    names, rows = [], []
    for cat, seqs in sequences.items():
        for i, s in enumerate(seqs):
            names.append(f"{cat}_{i}")
            rows.append(s)
    many = len(rows) > 1
    shifted = [("-" * ((i % 3) + (1 if (i == 0 and many) else 0))) + s for i, s in enumerate(rows)]
    if many:                                   # an internal gap in the template too
        t = shifted[0]
        shifted[0] = t[:3] + "-" + t[3:]
    width = max(len(s) for s in shifted)
    return names, [s + "-" * (width - len(s)) for s in shifted]


def fake_add_to_msa(msa, new_seq):
    """Toy profile alignment: new sequence first, everything right-padded to a common width."""
    width = max(len(new_seq), max(len(s) for s in msa))
    return [new_seq + "-" * (width - len(new_seq))] + [s + "-" * (width - len(s)) for s in msa]
