"""Pins oracle/draw.py: Philox known answers (Random123 kat vectors), exactness properties of
pg_exp, and the distribution tests the reference applies to generate_step
(/root/reference/test/test_esm_sampler.py:185-253)."""
import numpy as np

from oracle import draw


def test_philox_known_answers():
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, out in kat:
        got = draw.philox4x32_10(*[np.uint32(c) for c in ctr], key[0], key[1])
        assert tuple(int(g) for g in got) == out


def test_pg_exp_accuracy_and_edges():
    x = -np.abs(np.random.default_rng(0).standard_normal(100000).astype(np.float32) * 20)
    got = draw.pg_exp(x).astype(np.float64)
    want = np.exp(x.astype(np.float64))
    ok = want > 1e-37
    rel = np.abs(got[ok] / want[ok] - 1)
    assert rel.max() < 1e-5                      # |x|*2^-24 argument rounding dominates far from 0
    assert rel[x[ok] > -4].max() < 6e-7
    assert draw.pg_exp(np.float32(0.0)) == np.float32(1.0)
    assert draw.pg_exp(np.float32(-200.0)) == np.float32(0.0)
    assert (np.diff(draw.pg_exp(np.linspace(-30, 0, 5001).astype(np.float32))) >= 0).all()


def _counts(rows, n, **kw):
    cnt = {i: 0 for i in range(rows.shape[1])}
    toks = draw.draw_rows(np.repeat(rows, n, axis=0), kw.pop("valid_idx"), kw.pop("top_k", 0), kw.pop("sample", False),
                          None, np.arange(n), 0, np.zeros(n, dtype=np.int64), 0, 42)
    for t in toks:
        cnt[int(t)] += 1
    return cnt


def test_generate_step_without_idx_restriction():
    cnt = _counts(np.full((1, 6), .1, dtype=np.float32), 1000, valid_idx=list(range(6)))
    assert all(cnt[i] > 100 for i in range(6))


def test_generate_step_with_idx_restriction():
    cnt = _counts(np.full((1, 6), .1, dtype=np.float32), 1000, valid_idx=[1, 3, 5])
    assert cnt[0] == cnt[2] == cnt[4] == 0 and all(cnt[i] > 200 for i in (1, 3, 5))


def test_generate_step_with_idx_restriction_and_top_k():
    rows = np.array([[.4, .2, .4, .2, .1, .1]], dtype=np.float32)
    for valid in ([1, 3, 5], [3, 5, 1]):
        cnt = _counts(rows, 1000, valid_idx=valid, top_k=2)
        assert cnt[0] == cnt[2] == cnt[4] == cnt[5] == 0 and cnt[1] > 400 and cnt[3] > 400


def test_sample_flag_overrides_top_k():
    rows = np.array([[.4, .2, .4, .2, .1, .1]], dtype=np.float32)
    cnt = _counts(rows, 2000, valid_idx=[1, 3, 5], top_k=1, sample=True)
    assert cnt[5] > 300


def test_top1_is_argmax_lowest_index_on_ties():
    rows = np.array([[0, 3, 1, 3, 2, 3]], dtype=np.float32)
    assert draw.generate_step(rows, 0, top_k=1, valid_idx=[1, 3, 5]) == 1
    assert draw.generate_step(rows, 0, top_k=1, valid_idx=[5, 3, 1]) == 5


def test_distribution_matches_softmax():
    rng = np.random.default_rng(3)
    row = rng.standard_normal((1, 33)).astype(np.float32) * 2
    valid = list(range(4, 24))
    n = 200000
    toks = draw.draw_rows(np.repeat(row, n, axis=0), valid, 0, True, 0.8, np.arange(n) % 977, 5, np.arange(n) // 977, 1, 7)
    p = np.exp(row[0, valid].astype(np.float64) / 0.8)
    p /= p.sum()
    emp = np.bincount(toks, minlength=33)[valid] / n
    assert np.abs(emp - p).max() < 4 * np.sqrt(p.max() / n) + 1e-3
