"""Pins oracle/pyrandom.py against (a) the interpreter's own `random` module -- the module the
reference calls (esm_sampler.py:112,245; esm_msa_sampler.py:129,277) -- and (b) the committed
streams recorded next to the reference run (tests/golden/misc_ref.json)."""
import random

import pytest

from oracle.pyrandom import PyRandom
from _standin import load_json

SHAPES = [(25, 2), (256, 25), (257, 25), (512, 51), (10, 10), (21, 6), (22, 6), (85, 6), (86, 6), (5, 0), (1, 1)]


@pytest.mark.parametrize("seed", [0, 1, 7, 12345, 2**32 - 1, 2**32, 2**40 + 17, -5])
def test_matches_cpython_random(seed):
    r = PyRandom(seed)
    random.seed(seed)
    for n, k in SHAPES:
        assert r.sample(range(1, n + 1), k) == random.sample(range(1, n + 1), k)
        a, b = list(range(n)), list(range(n))
        r.shuffle(a)
        random.shuffle(b)
        assert a == b
        assert r.choices("abcdefg", k=5) == random.choices("abcdefg", k=5)
        assert r.random() == random.random()
    assert r.getstate() == random.getstate()


def test_state_interchange():
    random.seed(99)
    [random.random() for _ in range(1000)]
    r = PyRandom()
    r.setstate(random.getstate())
    assert r.sample(range(300), 30) == random.sample(range(300), 30)
    random.setstate(r.getstate())
    assert r.getrandbits(32) == random.getrandbits(32)


def test_committed_streams():
    for rec in load_json("misc_ref.json")["py_streams"]:
        r = PyRandom(rec["seed"])
        assert [r.sample(range(1, rec["n"] + 1), rec["k"]) for _ in range(3)] == rec["samples"]
        lst = list(range(rec["n"]))
        r.shuffle(lst)
        assert lst == rec["shuffled"]
        assert r.choices(["a", "b", "c"], k=5) == rec["choices"]
        assert r.getrandbits(32) == rec["next32"]


def test_sample_errors():
    with pytest.raises(ValueError):
        PyRandom(0).sample(range(3), 4)
