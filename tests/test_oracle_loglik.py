"""Pins the oracle's log-likelihood restatement (oracle/sampler.py) to the reference's own log_likelihood_batch
outputs recorded with the stand-in model (tests/golden/loglik.json)."""
import numpy as np

from oracle.sampler import esm_log_likelihood_batch, msa_log_likelihood_batch
from _standin import load_json, standin_logits_np

G = load_json("loglik.json")


def _kw(kw):
    kw = dict(kw)
    if "mask_distance" in kw and kw["mask_distance"] is None:
        kw["mask_distance"] = float("inf")
    return kw


def test_esm_log_likelihood_matches_reference():
    for c in G["esm"]:
        res = esm_log_likelihood_batch(standin_logits_np, c["seqs"], **_kw(c["kw"]))
        for (m, l), rm, rl in zip(res, c["means"], c["lists"]):
            assert len(l) == len(rl)
            assert np.abs(np.asarray(l) - np.asarray(rl)).max() < 2e-6
            assert abs(m - rm) < 2e-6


def test_msa_log_likelihood_matches_reference():
    for c in G["msa"]:
        res = msa_log_likelihood_batch(standin_logits_np, c["msas"], **_kw(c["kw"]))
        for (m, l), rm, rl in zip(res, c["means"], c["lists"]):
            assert len(l) == len(rl)
            assert np.abs(np.asarray(l) - np.asarray(rl)).max() < 2e-6
            assert abs(m - rm) < 2e-6
