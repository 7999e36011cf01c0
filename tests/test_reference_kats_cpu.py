"""Keeps the door to the reference's known-answer tests open: tests/golden/reference_kats.json -- the values the
checkpoint-gated replays compare the engine against (tests/test_gpu_esm1.py::test_reference_kats_with_the_pretrained_checkpoint,
tests/test_reference_suite_*.py) -- must hold EVERY float literal the reference's own tests assert at this boundary
(/root/reference/test/test_esm_sampler.py:269-340, /root/reference/test/test_esm_msa_sampler.py:248-397, 561-565).  The list of
those literals is tests/golden/reference_kat_literals.json, written by tests/golden/make_reference_kat_literals.py, which parses the
reference's test files with `ast` and imports nothing.  Where the reference is present (the build container) the list is re-derived
and compared, so neither file can drift."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _floats(o):
    if isinstance(o, float):
        yield o
    elif isinstance(o, dict):
        for v in o.values():
            yield from _floats(v)
    elif isinstance(o, (list, tuple)):
        for v in o:
            yield from _floats(v)


def test_reference_kats_json_holds_every_literal_the_reference_asserts():
    kats = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))
    lits = json.load(open(os.path.join(HERE, "golden", "reference_kat_literals.json")))["files"]
    have = {"test/test_esm_sampler.py": set(_floats(kats["esm6"])), "test/test_esm_msa_sampler.py": set(_floats(kats["msa1b"]))}
    n = 0
    for rel, per in lits.items():
        for func, vals in per.items():
            for v in vals:
                assert v in have[rel], "%s::%s asserts %r, which tests/golden/reference_kats.json does not hold" % (rel, func, v)
                n += 1
    assert n >= 60 and len(lits["test/test_esm_sampler.py"]) >= 6 and len(lits["test/test_esm_msa_sampler.py"]) >= 9
    # and nothing in the replay data that the reference does not assert (a typo would otherwise go unnoticed)
    ref_all = {rel: {v for vals in per.values() for v in vals} for rel, per in lits.items()}
    for rel, s in have.items():
        assert s <= ref_all[rel], "reference_kats.json holds values the reference's tests do not assert: %s" % sorted(s - ref_all[rel])


@pytest.mark.skipif(not os.path.isdir("/root/reference/test"), reason="the reference is only present in the build container")
def test_literal_list_is_what_the_reference_test_files_hold():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(HERE, "golden", "make_reference_kat_literals.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    assert mk.collect("/root/reference") == json.load(open(os.path.join(HERE, "golden", "reference_kat_literals.json")))["files"]
