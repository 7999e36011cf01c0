"""The integer/index ends of the iteration on the GPU -- mask scatter and the restrict / temperature /
top-k / categorical draw + write-back kernels -- against the CPU oracle (bit-exact given identical
logits) and through the distribution tests the reference applies to generate_step
(/root/reference/test/test_esm_sampler.py:185-253)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import draw as odraw
from protein_gibbs_sampler_amd import _lib
from protein_gibbs_sampler_amd.esm_sampler import generate_step

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _run_sample(tokens, logits, idx, params, it, row_map=None):
    L = _lib.lib()
    d_tok = torch.from_numpy(tokens).to(DEV)
    d_log = torch.from_numpy(logits).to(DEV)
    d_idx = torch.from_numpy(idx).to(DEV)
    d_map = torch.from_numpy(row_map).to(DEV) if row_map is not None else None
    d_out = torch.full(idx.shape, -7, dtype=torch.int32, device=DEV)
    n_rows, width = tokens.shape
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.pg_sample_writeback_device(stream, _p(d_tok), n_rows, width, _p(d_log), logits.shape[-1], _p(d_idx),
                                            _p(d_map) if d_map is not None else None, idx.shape[0], idx.shape[1],
                                            ctypes.byref(params), it, _p(d_out)))
    torch.cuda.synchronize()
    return d_tok.cpu().numpy(), d_out.cpu().numpy()


@pytest.mark.parametrize("top_k,burnin,temp,valid", [(0, float("inf"), None, list(range(4, 24))),
                                                     (3, 0, 0.7, list(range(4, 24))),
                                                     (1, 0, None, list(range(4, 24)) + [30]),
                                                     (5, 2, 1.3, [9, 4, 30, 17, 5]),
                                                     (40, 0, None, list(range(4, 24)))])
def test_draw_bit_exact_vs_oracle(top_k, burnin, temp, valid):
    rng = np.random.default_rng(7)
    n_rows, width, V, P = 37, 50, 33, 6
    logits = (rng.standard_normal((n_rows, width, V)) * 3).astype(np.float32)
    logits[3, :, 5] = logits[3, :, 9]                    # exact ties
    tokens = rng.integers(4, 24, (n_rows, width)).astype(np.int32)
    idx = np.stack([rng.choice(np.arange(1, width), P, replace=False) for _ in range(n_rows)]).astype(np.int32)
    idx[5, 2] = -1                                       # ragged padding
    idx[6, 1] |= 1 << 30                                 # shadowed slot: drawn, not written
    for it in (0, 1, 2, 5):
        params = _lib.make_sample_params(True, 32, top_k, burnin, temp, valid, rng_seed=0xABCDEF12345, rng_stream=3,
                                         row_id_base=1000, iter_base=10)
        new_tok, picked = _run_sample(tokens, logits, idx, params, it)
        pos = idx & 0x3FFFFFFF
        ok = idx >= 0
        rows = logits[np.arange(n_rows)[:, None], np.where(ok, pos, 0)].reshape(-1, V)
        want = odraw.draw_rows(rows, valid, top_k, it < burnin, temp, 1000 + np.repeat(np.arange(n_rows), P), 10 + it,
                               np.tile(np.arange(P), n_rows), 3, 0xABCDEF12345).reshape(n_rows, P)
        assert (picked[ok] == want[ok]).all()
        assert (picked[~ok] == -1).all()
        exp = tokens.copy()
        for r in range(n_rows):
            for p in range(P):
                if idx[r, p] >= 0 and not (idx[r, p] & (1 << 30)):
                    exp[r, pos[r, p]] = want[r, p]
        assert (new_tok == exp).all()


def test_row_map_and_mask_scatter():
    L = _lib.lib()
    rng = np.random.default_rng(1)
    tokens = rng.integers(4, 24, (6, 20)).astype(np.int32)
    idx = np.array([[3, 7, -1], [1, 2, 19]], dtype=np.int32)
    row_map = np.array([5, 2], dtype=np.int32)
    d_tok, d_idx, d_map = (torch.from_numpy(a).to(DEV) for a in (tokens, idx, row_map))
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.pg_mask_scatter_device(stream, _p(d_tok), 6, 20, _p(d_idx), _p(d_map), 2, 3, 32))
    torch.cuda.synchronize()
    exp = tokens.copy()
    exp[5, [3, 7]] = 32
    exp[2, [1, 2, 19]] = 32
    assert (d_tok.cpu().numpy() == exp).all()
    logits = (rng.standard_normal((6, 20, 33)) * 2).astype(np.float32)
    params = _lib.make_sample_params(True, 32, 1, 0, None, list(range(4, 24)), 0)
    new_tok, picked = _run_sample(tokens, logits, idx, params, 0, row_map=row_map)
    assert picked[0, 0] == 4 + logits[5, 3, 4:24].argmax() and picked[1, 2] == 4 + logits[2, 19, 4:24].argmax()
    assert new_tok[5, 3] == picked[0, 0] and new_tok[2, 19] == picked[1, 2] and picked[0, 2] == -1


# ---- the reference's own generate_step tests, through the HIP kernel ---------------------------------
def _counts(out, n=1000, **kw):
    cnts = {i: 0 for i in range(out.shape[1])}
    for _ in range(n):
        cnts[generate_step(out, 0, **kw).item()] += 1
    return cnts


def test_generate_step_without_idx_restriction():
    c = _counts(torch.tensor([[.1, .1, .1, .1, .1, .1]]))
    assert all(c[i] > 100 for i in range(6))


def test_generate_step_with_idx_restriction():
    c = _counts(torch.tensor([[.1, .1, .1, .1, .1, .1]]), valid_idx=[1, 3, 5])
    assert c[0] == c[2] == c[4] == 0 and c[1] > 200 and c[3] > 200 and c[5] > 200


@pytest.mark.parametrize("valid", [[1, 3, 5], [3, 5, 1]])
def test_generate_step_with_idx_restriction_and_top_k(valid):
    c = _counts(torch.tensor([[.4, .2, .4, .2, .1, .1]]), top_k=2, valid_idx=valid)
    assert c[0] == c[2] == c[4] == c[5] == 0 and c[1] > 400 and c[3] > 400
