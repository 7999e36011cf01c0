"""The reference's own test file for this path, /root/reference/test/test_esm_sampler.py, test by test and under the same names,
against this package's drop-in `ESM_sampler` (the line each test restates is in its docstring).

The reference loads `models.ESM6()` (the pretrained 43 M ESM-1 model) on the CPU.  Here the host-side tests run anywhere on an
ESM6 holder with seeded synthetic weights; everything that needs a forward pass runs on the MI355X (`-m gpu`; the product has no
CPU forward).  The reference's numeric known answers (log-likelihoods under the pretrained weights) are held as data in
tests/golden/reference_kats.json: with `esm1_t6_43M_UR50S.pt` in torch's hub cache they are compared directly (strict mode,
2e-3), without it the same calls are checked through the properties the reference's tables encode (score == mean of the
per-position scores; a mask distance >= the sequence length == one position masked at a time; batch size does not matter;
the batch call == the single calls)."""
import os
import warnings
from statistics import mean

import pytest
import torch

from protein_gibbs_sampler_amd import esm_sampler, models
from protein_gibbs_sampler_amd.esm_sampler import ESM_ALLOWED_AMINO_ACIDS, generate_step
from _standin import load_json

KAT = load_json("reference_kats.json")["esm6"]
_CKPT = os.path.expanduser("~/.cache/torch/hub/checkpoints/esm1_t6_43M_UR50S.pt")
PRETRAINED = os.path.exists(_CKPT)
gpu = pytest.mark.gpu


def _esm6(precision="fp32"):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return models.ESM6(precision=precision) if PRETRAINED else models.ESM6(synthetic=True, precision=precision)


@pytest.fixture(scope="module")
def esm6():
    return _esm6()


@pytest.fixture(scope="module")
def esm_sampler_fixture(esm6):
    """Host-side sampler (the reference's fixture, test_esm_sampler.py:15-18)."""
    return esm_sampler.ESM_sampler(esm6, device="cpu")


@pytest.fixture(scope="module")
def gpu_sampler(esm6):
    return esm_sampler.ESM_sampler(esm6, device="gpu")


# ---- construction (test_esm_sampler.py:24-40) --------------------------------------------------------------------------------------
def test_sampler_init_cpu(esm6):
    """:24-27"""
    esm_sampler.ESM_sampler(esm6, device="cpu")


@gpu
def test_sampler_init_gpu(esm6):
    """:29-32"""
    assert esm_sampler.ESM_sampler(esm6, device="gpu").cuda


@gpu
def test_sampler_init_cuda0(esm6):
    """:34-37"""
    assert esm_sampler.ESM_sampler(esm6, device="cuda:0").cuda


def test_sampler_init_gpu_when_not_available(esm6, mock_no_gpu):
    """:39-40"""
    pytest.raises(Exception, esm_sampler.ESM_sampler, esm6, device="gpu")


# ---- seeds -> token batches, ESM-1 alphabet: <cls> = 32, <mask> = 33, no <eos> (:43-88) -------------------------------------------
def test_get_init_seq_empty(esm_sampler_fixture):
    """:43-47"""
    assert esm_sampler_fixture.get_init_seq("", 5, 1).tolist() == [[32, 33, 33, 33, 33, 33]]


def test_get_init_seq_string_seed(esm_sampler_fixture):
    """:50-54"""
    assert esm_sampler_fixture.get_init_seq("AA", 5, 1).tolist() == [[32, 5, 5, 33, 33, 33]]


def test_get_init_seq_string_seed_lowercase(esm_sampler_fixture):
    """:57-61"""
    assert esm_sampler_fixture.get_init_seq("aa", 5, 1).tolist() == [[32, 5, 5, 33, 33, 33]]


def test_get_init_seq_string_fails_if_non_standard_supplied(esm_sampler_fixture):
    """:64-70"""
    with pytest.raises(Exception) as e:
        esm_sampler_fixture.get_init_seq("X", 5, 1)
    assert str(e.value) == "Invalid input character: X"


def test_get_init_seq_array_of_seeds(esm_sampler_fixture):
    """:73-77"""
    assert esm_sampler_fixture.get_init_seq(["Aa"], 5, 1).tolist() == [[32, 5, 5, 33, 33, 33]]


def test_get_init_seq_array_of_seeds_builds_batch_randomly(esm_sampler_fixture):
    """:80-88"""
    out = esm_sampler_fixture.get_init_seq(["AA", "A"], 5, 3)
    assert len(out) == 3
    for item in out.tolist():
        assert item in ([32, 5, 5, 33, 33, 33], [32, 5, 33, 33, 33, 33])


# ---- generate: shapes of what comes back (:90-124, :256-261) -----------------------------------------------------------------------
@gpu
def test_generate_batch_equals_seqs(gpu_sampler):
    """:90-96"""
    out = gpu_sampler.generate(4, "", batch_size=4, max_len=10, show_progress_bar=False)
    assert len(out) == 4 and all(len(s) == 10 for s in out)


@gpu
@pytest.mark.parametrize("batch_size, num_positions,mask,leader_length,in_order",
                         [(3, 1, True, 1, True), (3, 1, False, 1, True), (3, 1, True, 1, False), (3, 1, False, 1, False),
                          (3, 1, True, -1, False), (10, 3, False, 1, False)])
def test_generate_batch_with_varying_input(gpu_sampler, batch_size, num_positions, mask, leader_length, in_order):
    """:99-115"""
    out = gpu_sampler.generate(4, "AAAAAAAAAA", batch_size=batch_size, max_len=10, num_iters=2, num_positions=num_positions,
                               mask=mask, leader_length=leader_length, in_order=in_order, show_progress_bar=False)
    assert len(out) == 4 and all(len(s) == 10 for s in out)


@gpu
def test_generate_batch_greater_than_seqs(gpu_sampler):
    """:118-124"""
    out = gpu_sampler.generate(4, "", batch_size=10, max_len=10, show_progress_bar=False)
    assert len(out) == 4 and all(len(s) == 10 for s in out)


@gpu
def test_generate_batch_only_includes_allowed_aa(gpu_sampler):
    """:256-261"""
    out = gpu_sampler.generate(10, "", batch_size=10, max_len=25, show_progress_bar=False)
    assert len(out) == 10
    for sequence in out:
        assert not set(sequence) - set(ESM_ALLOWED_AMINO_ACIDS)


# ---- index helpers and the mask scatter (:130-163) ----------------------------------------------------------------------------------
def test_get_target_index_in_order(esm_sampler_fixture):
    """:130-137"""
    last_i, target_indexes = esm_sampler_fixture.get_target_index_in_order(batch_size=2, indexes=[0, 1, 2, 3], next_i=1, num_positions=2)
    assert len(target_indexes) == 2 and last_i == 3 and target_indexes == [[2, 3], [2, 3]]


def test_get_target_index_randomly(esm_sampler_fixture):
    """:140-149"""
    indexes = [0, 1, 2, 3]
    target_indexes = esm_sampler_fixture.get_random_target_index(batch_size=2, indexes=indexes, num_positions=3)
    assert len(target_indexes) == 2 and len(target_indexes[0]) == 3
    assert all(item in indexes for item in target_indexes[0])


def test_mask_indexes(esm_sampler_fixture):
    """:152-163"""
    batch = [[1, 1, 1, 1], [1, 1, 1, 1], [1, 1, 1, 1]]
    esm_sampler_fixture.mask_target_indexes(batch, [[2, 3], [1, 2], [0, 1]])
    assert batch == [[1, 1, 33, 33], [1, 33, 33, 1], [33, 33, 1, 1]]


# ---- the tokens a draw may produce (:166-182) ----------------------------------------------------------------------------------------
def _allowed_toks(s):
    return {s.model.alphabet.get_tok(idx) for idx in s.valid_aa_idx}


def test_allowable_amino_acid_locations_only_contain_standard_aa(esm_sampler_fixture):
    """:170-175"""
    allowed = _allowed_toks(esm_sampler_fixture)
    assert allowed.issubset(set(esm_sampler_fixture.model.alphabet.standard_toks)) and allowed == set(ESM_ALLOWED_AMINO_ACIDS)


def test_allowable_amino_acid_locations_do_not_contain_amino_acids_we_cant_create(esm_sampler_fixture):
    """:178-182"""
    assert _allowed_toks(esm_sampler_fixture).isdisjoint(set("XBUXZO.-"))


# ---- generate_step: 1000 draws per case through the HIP draw kernel (:185-253) ------------------------------------------------------
def _counts(probs, **kw):
    cnts = {idx: 0 for idx in range(6)}
    for _ in range(1000):
        cnts[generate_step(torch.tensor([probs]), 0, **kw).item()] += 1
    return cnts


@gpu
def test_generate_step_without_idx_restriction():
    """:185-199"""
    cnts = _counts([.1, .1, .1, .1, .1, .1])
    assert all(cnts[i] > 100 for i in range(6))


@gpu
def test_generate_step_with_idx_restriction():
    """:202-217"""
    cnts = _counts([.1, .1, .1, .1, .1, .1], valid_idx=[1, 3, 5])
    assert cnts[0] == cnts[2] == cnts[4] == 0 and cnts[1] > 200 and cnts[3] > 200 and cnts[5] > 200


@gpu
def test_generate_step_with_idx_restriction_and_top_k():
    """:220-235 (index 5 is valid but outside the top two)"""
    cnts = _counts([.4, .2, .4, .2, .1, .1], top_k=2, valid_idx=[1, 3, 5])
    assert cnts[0] == cnts[2] == cnts[4] == cnts[5] == 0 and cnts[1] > 400 and cnts[3] > 400


@gpu
def test_generate_step_with_out_of_order_idx_restriction():
    """:238-253"""
    cnts = _counts([.4, .2, .4, .2, .1, .1], top_k=2, valid_idx=[3, 5, 1])
    assert cnts[0] == cnts[2] == cnts[4] == cnts[5] == 0 and cnts[1] > 400 and cnts[3] > 400


# ---- log-likelihoods (:269-340) ------------------------------------------------------------------------------------------------------
def _check(value, per_position, expected):
    assert value == pytest.approx(mean(per_position), abs=1e-5)
    if PRETRAINED:
        assert value == pytest.approx(expected, abs=2e-3)


@gpu
def test_log_likelihood_with_mask(gpu_sampler):
    """:269-280"""
    for seq, v in zip(KAT["seqs"], KAT["with_mask"]):
        seq_prob, pos_probs = gpu_sampler.log_likelihood(seq)
        assert len(pos_probs) == len(seq)
        _check(seq_prob, pos_probs, v)


@gpu
def test_log_likelihood_without_mask(gpu_sampler):
    """:282-286"""
    for seq, v in zip(KAT["seqs"], KAT["without_mask"]):
        seq_prob, pos_probs = gpu_sampler.log_likelihood(seq, with_masking=False)
        _check(seq_prob, pos_probs, v)
        # seeing the residue can only help on average: the unmasked score of a sequence is the larger one (as in the reference's tables)
        if PRETRAINED:
            assert seq_prob > gpu_sampler.log_likelihood(seq)[0]


@gpu
def test_log_likelihood_batch_with_mask(gpu_sampler):
    """:288-297"""
    results = list(gpu_sampler.log_likelihood_batch(KAT["seqs"], with_masking=True))
    assert len(results) == 3
    for (v, per), seq, want in zip(results, KAT["seqs"], KAT["with_mask"]):
        _check(v, per, want)
        assert v == pytest.approx(gpu_sampler.log_likelihood(seq)[0], abs=1e-4)            # the batch call == the single call


@gpu
def test_log_likelihood_batch_without_mask(gpu_sampler):
    """:300-308"""
    results = list(gpu_sampler.log_likelihood_batch(KAT["seqs"], with_masking=False))
    assert len(results) == 3
    for (v, per), want in zip(results, KAT["without_mask"]):
        _check(v, per, want)


@gpu
@pytest.mark.parametrize("mask_distance", [1, 2, 5, 10, 20, 40])
def test_likelihood_batch_with_individual_masking_distance(gpu_sampler, mask_distance):
    """:311-326: every `mask_distance`-th position masked per forward; from the sequence's own length on (20 for input 2, 36 for
    input 1) that is one position at a time, i.e. the with_mask value"""
    seqs = KAT["seqs"][:2]
    actual = list(gpu_sampler.log_likelihood_batch(seqs, with_masking=True, mask_distance=mask_distance))
    for i in range(2):
        _check(actual[i][0], actual[i][1], KAT["mask_distance"][str(mask_distance)][i])
        if mask_distance >= len(seqs[i]):
            assert actual[i][0] == pytest.approx(gpu_sampler.log_likelihood(seqs[i])[0], abs=1e-4)


@gpu
@pytest.mark.parametrize("batch_size", [1, 2, 5, 100])
@pytest.mark.parametrize("mask_distance", [1, 2, 5])
def test_likelihood_batch_handles_batch_sizes(gpu_sampler, batch_size, mask_distance):
    """:328-340: the forward batch size changes nothing"""
    seqs = KAT["seqs"][:2]
    actual = list(gpu_sampler.log_likelihood_batch(seqs, with_masking=True, mask_distance=mask_distance, batch_size=batch_size))
    base = list(gpu_sampler.log_likelihood_batch(seqs, with_masking=True, mask_distance=mask_distance))
    for i in range(2):
        _check(actual[i][0], actual[i][1], KAT["mask_distance"][str(mask_distance)][i])
        assert actual[i][0] == pytest.approx(base[i][0], abs=1e-4)
