"""Pins oracle/sampler.py (the CPU restatement of the Gibbs control loops) against the I/O pairs
recorded from the reference itself (tests/golden/sampler_*.json, made by make_golden.py)."""
import numpy as np
import pytest

from oracle.pyrandom import PyRandom
from oracle.sampler import OracleESMSampler, OracleMSASampler, partition, clean_seed_seq, ESM_ALLOWED
from _standin import load_json, standin_logits_np

ESM = load_json("sampler_esm.json")
MSA = load_json("sampler_msa.json")
MISC = load_json("misc_ref.json")


def _kw(kw):
    kw = dict(kw)
    if kw.get("burnin") is None and "burnin" in kw:
        kw["burnin"] = float("inf")
    return kw


@pytest.mark.parametrize("name", sorted(ESM))
def test_esm_generate_matches_reference(name):
    c = ESM[name]
    rng = PyRandom(c["pyseed"])
    s = OracleESMSampler(standin_logits_np, rng=rng)
    strings = s.generate(c["n_samples"], c["seed_seq"], **_kw(c["kw"]))
    assert s.trace["targets"] == c["targets"]                       # bit-exact position selection
    assert len(s.trace["forward_inputs"]) == len(c["forward_inputs"])
    mask = 32
    deterministic = c["kw"].get("burnin", None) == 0 and c["kw"].get("top_k") == 1
    for mine, ref in zip(s.trace["forward_inputs"], c["forward_inputs"]):
        ref = np.asarray(ref)
        assert mine.shape == ref.shape
        if deterministic:
            assert (mine == ref).all()                              # mask scatter + argmax write-back
        else:
            assert ((mine == mask) == (ref == mask)).all()               # mask pattern depends on the targets only
    assert (s.trace["forward_inputs"][0] == np.asarray(c["forward_inputs"][0])).all()
    if deterministic:
        assert strings == c["strings"]
    else:
        assert len(strings) == len(c["strings"])
    assert rng.getrandbits(32) == c["py_state_after"][-1]           # same RNG consumption


@pytest.mark.parametrize("name", sorted(k for k in MSA if not k.startswith("single")))
def test_msa_generate_matches_reference(name):
    c = MSA[name]
    rng = PyRandom(c["pyseed"])
    s = OracleMSASampler(standin_logits_np, rng=rng)
    strings = s.generate(c["n_samples"], c["seed_msa"], **_kw(c["kw"]))
    assert s.trace["targets"] == c["targets"]
    for mine, ref in zip(s.trace["forward_inputs"], c["forward_inputs"]):
        assert (mine == np.asarray(ref)).all()
    assert strings == c["strings"]
    assert rng.getrandbits(32) == c["py_state_after"][-1]


@pytest.mark.parametrize("name", sorted(k for k in MSA if k.startswith("single")))
def test_msa_generate_single_matches_reference(name):
    c = MSA[name]
    rng = PyRandom(c["pyseed"])
    s = OracleMSASampler(standin_logits_np, rng=rng)
    string = s.generate_single(list(c["seed_msa"]), **c["kw"])
    assert len(s.trace["forward_inputs"]) == len(c["forward_inputs"])
    for mine, ref in zip(s.trace["forward_inputs"], c["forward_inputs"]):
        assert (mine == np.asarray(ref)).all()
    assert string == c["string"]
    assert rng.getrandbits(32) == c["py_state_after"][-1]


def test_partition_matches_reference():
    for rec in MISC["partition"]:
        assert partition(list(range(1, rec["n"] + 1)), rec["parts"]) == rec["out"]
    assert MISC["partition_empty"] == "ZeroDivisionError"
    with pytest.raises(ZeroDivisionError):
        partition([], 3)


def test_partition_reference_kats():
    """/root/reference/test/test_esm_msa_sampler.py:538-557."""
    ten = list(range(1, 11))
    assert partition(ten, 3) == [[1, 2, 3, 4], [5, 6, 7], [8, 9, 10]]
    assert partition(ten, 4) == [[1, 2, 3], [4, 5, 6], [7, 8], [9, 10]]
    assert partition(ten, 600) == [[i] for i in ten]


def test_clean_seed_errors():
    for s, bad in MISC["esm_clean_errors"].items():
        if bad is None:
            clean_seed_seq(s, ESM_ALLOWED)
        else:
            with pytest.raises(Exception) as e:
                clean_seed_seq(s, ESM_ALLOWED)
            assert sorted(str(e.value)[len("Invalid input character: "):].split(",")) == bad
