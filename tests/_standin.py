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


def make_standin_torch_module():
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
            return {"logits": logits}

    return Net()


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)
