"""The reference's own test file for the MSA path, /root/reference/test/test_esm_msa_sampler.py, test by test and under the same
names, against this package's drop-in `ESM_MSA_sampler` (the lines each test restates are in its docstring).

The reference loads `models.ESM_MSA1()` (pretrained esm_msa1b_t12_100M_UR50S) on the CPU.  Here the host-side tests run anywhere on
a holder with seeded synthetic weights; everything that needs a forward pass runs on the MI355X (`-m gpu`).  The reference's numeric
known answers are data in tests/golden/reference_kats.json: compared directly (strict mode, 2e-3) when the checkpoint is in torch's
hub cache, otherwise the same calls are checked through the properties the reference's tables encode (score == mean of the
per-position scores, batch call == single calls, batch size does not matter, a mask distance >= the width == one column at a time)."""
import copy
import os
import warnings
from io import StringIO
from statistics import mean

import pytest

from protein_gibbs_sampler_amd import esm_msa_sampler, likelihood_esm_msa, models, msa_tools
from protein_gibbs_sampler_amd.esm_msa_sampler import ESM_MSA_ALLOWED_AMINO_ACIDS
from _standin import fake_add_to_msa, fake_run_phmmer, load_json

KAT = load_json("reference_kats.json")["msa1b"]
_CKPT = os.path.expanduser("~/.cache/torch/hub/checkpoints/esm_msa1b_t12_100M_UR50S.pt")
PRETRAINED = os.path.exists(_CKPT)
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def esm_msa():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return models.ESM_MSA1(precision="fp32") if PRETRAINED else models.ESM_MSA1(synthetic=True, precision="fp32")


@pytest.fixture(scope="module")
def msa_sampler(esm_msa):
    """Host-side sampler (the reference's fixture, test_esm_msa_sampler.py:20-23)."""
    return esm_msa_sampler.ESM_MSA_sampler(esm_msa, device="cpu")


@pytest.fixture(scope="module")
def gpu_sampler(esm_msa):
    return esm_msa_sampler.ESM_MSA_sampler(esm_msa, device="gpu")


@pytest.fixture()
def msa_batch_example():
    """:248-262"""
    return copy.deepcopy(KAT["msas"])


# ---- construction (:27-40) ----------------------------------------------------------------------------------------------------------
def test_sampler_init_cpu(esm_msa):
    """:27-30"""
    esm_msa_sampler.ESM_MSA_sampler(esm_msa, device="cpu")


@gpu
def test_sampler_init_gpu(esm_msa):
    """:32-35"""
    assert esm_msa_sampler.ESM_MSA_sampler(esm_msa, device="gpu").cuda


@gpu
def test_sampler_init_cuda0(esm_msa):
    """:37-40"""
    assert esm_msa_sampler.ESM_MSA_sampler(esm_msa, device="cuda:0").cuda


# ---- tokens <-> strings (:43-84) -----------------------------------------------------------------------------------------------------
def test_untokenize_batch(msa_sampler):
    """:43-52"""
    batch = [[[0, 5, 5, 5, 32, 32], [0, 5, 25, 25, 32, 32]], [[0, 5, 25, 23, 13, 32]]]
    assert msa_sampler.untokenize_batch(batch) == ["AAA<mask><mask>", "ABB<mask><mask>", "ABCD<mask>"]


def test_get_init_msa(msa_sampler):
    """:55-66"""
    seed = ["AAA", "ACC", "ACDE"]
    result = msa_sampler.get_init_msa(seed, 5, 2)
    assert tuple(result.shape) == (2, len(seed), 5 + 1)
    assert result[0][0].tolist() == [0, 5, 5, 5, 32, 32]
    assert result[0][1].tolist() == [0, 5, 23, 23, 32, 32]
    assert result[0][2].tolist() == [0, 5, 23, 13, 9, 32]


def test_get_init_msa_lowercase(msa_sampler):
    """:69-75"""
    result = msa_sampler.get_init_msa(["aaa", "aCC", "aCDE"], 5, 2)
    assert result[0].tolist() == [[0, 5, 5, 5, 32, 32], [0, 5, 23, 23, 32, 32], [0, 5, 23, 13, 9, 32]]


def test_get_init_msa_fails_if_non_standard_supplied(msa_sampler):
    """:78-84"""
    with pytest.raises(Exception) as e:
        msa_sampler.get_init_msa(["X"], 2)
    assert str(e.value) == "Invalid input character: X"


# ---- generate (:87-129, :240-245) ----------------------------------------------------------------------------------------------------
@gpu
def test_generate_batch_equals_seqs(gpu_sampler):
    """:87-92"""
    out = gpu_sampler.generate(4, ["AAA", "AAC"], batch_size=4, max_len=3, show_progress_bar=False)
    assert len(out) == 4 and all(len(s) == 3 for s in out)


@gpu
@pytest.mark.parametrize("batch_size, num_positions,mask,leader_length,in_order",
                         [(3, 1, True, 1, True), (3, 1, False, 1, True), (3, 1, True, 1, False), (3, 1, False, 1, False),
                          (3, 1, True, -1, False), (10, 3, False, 1, False)])
def test_generate_batch_with_varying_input(gpu_sampler, batch_size, num_positions, mask, leader_length, in_order):
    """:95-110"""
    out = gpu_sampler.generate(4, ["AAA", "AAC"], batch_size=batch_size, max_len=3, num_iters=2, num_positions=num_positions,
                               mask=mask, leader_length=leader_length, in_order=in_order, show_progress_bar=False)
    assert len(out) == 4 and all(len(s) == 3 for s in out)


@gpu
def test_generate_batch_single_iteration(gpu_sampler):
    """:113-122: one in-order iteration resamples column 1 only, so the seeds' second and third residues come back untouched"""
    out = gpu_sampler.generate(4, ["AAA", "AAC"], num_iters=1, max_len=5, num_positions=1, in_order=True, show_progress_bar=False)
    assert len(out) == 4
    assert [s[1:3] for s in out] == ["AA", "AC", "AA", "AC"]


@gpu
def test_generate_batch_randomly(gpu_sampler):
    """:125-129"""
    out = gpu_sampler.generate(4, ["AAA", "AAC"], num_iters=1, max_len=5, num_positions=1, in_order=False, show_progress_bar=False)
    assert len(out) == 4


@gpu
def test_generate_batch_only_includes_allowed_aa(gpu_sampler):
    """:240-245"""
    out = gpu_sampler.generate(10, ["AAA", "AAC"], num_iters=1, max_len=25, show_progress_bar=False)
    assert len(out) == 10
    for sequence in out:
        assert not set(sequence) - set(ESM_MSA_ALLOWED_AMINO_ACIDS)


# ---- index helpers and the mask scatter (:132-218) -----------------------------------------------------------------------------------
def test_get_target_indexes_in_order(msa_sampler):
    """:132-141"""
    last_i, target_indexes = msa_sampler.get_target_index_in_order(batch_size=2, indexes=[0, 1, 2, 3], next_i=1, num_positions=2,
                                                                   num_sequences=3)
    assert last_i == 3
    assert target_indexes == [[[2, 3], [2, 3], [2, 3]], [[2, 3], [2, 3], [2, 3]]]


def test_get_target_indexes_randomly(msa_sampler):
    """:144-155"""
    indexes = [0, 1, 2, 3]
    target_indexes = msa_sampler.get_random_target_index(batch_size=2, indexes=indexes, num_positions=2, num_sequences=3)
    assert len(target_indexes) == 2 and len(target_indexes[0]) == 3 and len(target_indexes[0][0]) == 2
    assert all(item in indexes for item in target_indexes[0][0])


def test_get_target_indexes_all_positions(msa_sampler):
    """:158-165"""
    target_indexes = msa_sampler.get_target_indexes_all_positions(batch_size=2, indexes=[0, 1, 2, 3], num_sequences=3)
    assert target_indexes == [[[0, 1, 2, 3]] * 3] * 2


def test_mask_indexes(msa_sampler):
    """:168-182"""
    batch = [[[1, 1, 1, 1], [1, 1, 1, 1], [1, 1, 1, 1]], [[1, 1, 1, 1], [1, 1, 1, 1], [1, 1, 1, 1]]]
    msa_sampler.mask_target_indexes(batch, [[[2, 3], [1, 2], [0, 1]], [[0, 1], [2, 1], [3, 2]]])
    assert batch == [[[1, 1, 32, 32], [1, 32, 32, 1], [32, 32, 1, 1]], [[32, 32, 1, 1], [1, 32, 32, 1], [1, 1, 32, 32]]]


def test_calculate_indexes_no_rollover(msa_sampler):
    """:185-194"""
    assert msa_sampler.calculate_indexes(None, 1, 5, False) == ([2, 3, 4, 5], 0)


def test_calculate_indexes_with_rollover(msa_sampler):
    """:197-206"""
    assert msa_sampler.calculate_indexes(None, 1, 5, True) == ([1, 2, 3, 4, 5], -1)


def test_calculate_indexes_when_indexes_supplied(msa_sampler):
    """:209-218"""
    assert msa_sampler.calculate_indexes([2, 3, 4, 5], 1, 5, False) == ([2, 3, 4, 5], -1)


# ---- the tokens a draw may produce (:221-237) ---------------------------------------------------------------------------------------
def _allowed_toks(s):
    return {s.model.alphabet.get_tok(idx) for idx in s.valid_aa_idx}


def test_allowable_amino_acid_locations_only_contain_standard_aa(msa_sampler):
    """:225-230"""
    allowed = _allowed_toks(msa_sampler)
    assert allowed.issubset(set(msa_sampler.model.alphabet.standard_toks)) and allowed == set(ESM_MSA_ALLOWED_AMINO_ACIDS)


def test_allowable_amino_acid_locations_do_not_contain_amino_acids_we_cant_create(msa_sampler):
    """:233-237"""
    assert _allowed_toks(msa_sampler).isdisjoint(set("XBUXZO."))


# ---- log-likelihoods of the first row of an alignment (:265-358) ---------------------------------------------------------------------
def _check(value, per_position, expected):
    assert mean(per_position) == pytest.approx(value, abs=1e-5)
    if PRETRAINED:
        assert value == pytest.approx(expected, abs=2e-3)


@gpu
def test_likelihood_without_masking(gpu_sampler, msa_batch_example):
    """:265-272"""
    for msa, v in zip(msa_batch_example, KAT["without_mask"]):
        seq_prob, pos_probs = gpu_sampler.log_likelihood(msa, target_index=0, with_masking=False)
        assert len(pos_probs) == len(msa[0])
        _check(seq_prob, pos_probs, v)


@gpu
def test_likelihood_with_individual_masking(gpu_sampler, msa_batch_example):
    """:275-282"""
    for msa, v in zip(msa_batch_example, KAT["with_mask"]):
        seq_prob, pos_probs = gpu_sampler.log_likelihood(msa, target_index=0, with_masking=True)
        _check(seq_prob, pos_probs, v)


@gpu
def test_likelihood_with_masking_entire_sequence(gpu_sampler, msa_batch_example):
    """:285-292"""
    for msa, v in zip(msa_batch_example, KAT["mask_distance"]["1"]):
        seq_prob, pos_probs = gpu_sampler.log_likelihood(msa, target_index=0, with_masking=True, mask_distance=1)
        _check(seq_prob, pos_probs, v)


@gpu
def test_likelihood_with_masking_entire_sequence_skip_gap(gpu_sampler, msa_batch_example):
    """:295-300: a gap in the scored row is left out of the mean unless count_gaps"""
    msa_batch_example[0][0] = KAT["skip_gap_first_row"]
    seq_prob, pos_probs = gpu_sampler.log_likelihood(msa_batch_example[0], target_index=0, with_masking=True, mask_distance=1,
                                                     count_gaps=False)
    assert len(pos_probs) == len(KAT["skip_gap_first_row"]) - 1
    _check(seq_prob, pos_probs, KAT["skip_gap_value"])


@gpu
def test_likelihood_batch_without_masking(gpu_sampler, msa_batch_example):
    """:302-307 (a ragged list: 28 and 44 columns in one padded tensor, as the reference scores it)"""
    result = list(gpu_sampler.log_likelihood_batch(msa_batch_example, target_index=0, with_masking=False))
    assert len(result) == 2
    for (v, per), want in zip(result, KAT["without_mask"]):
        _check(v, per, want)


@gpu
def test_likelihood_batch_with_individual_masking(gpu_sampler, msa_batch_example):
    """:309-314"""
    result = list(gpu_sampler.log_likelihood_batch(msa_batch_example, target_index=0, with_masking=True))
    for (v, per), msa, want in zip(result, msa_batch_example, KAT["with_mask"]):
        _check(v, per, want)
        assert v == pytest.approx(gpu_sampler.log_likelihood(msa, target_index=0, with_masking=True)[0], abs=1e-4)


@gpu
def test_likelihood_batch_with_masking_entire_sequence(gpu_sampler, msa_batch_example):
    """:316-322"""
    result = list(gpu_sampler.log_likelihood_batch(msa_batch_example, target_index=0, with_masking=True, mask_distance=1))
    for (v, per), want in zip(result, KAT["mask_distance"]["1"]):
        _check(v, per, want)


@gpu
@pytest.mark.parametrize("mask_distance", [1, 2, 5, 10, 28, 50])
def test_likelihood_batch_with_individual_masking_distance(gpu_sampler, msa_batch_example, mask_distance):
    """:325-339: from the alignment's own width on (28 columns for input 1, 44 for input 2) a mask distance is one column at a time"""
    result = list(gpu_sampler.log_likelihood_batch(msa_batch_example, target_index=0, with_masking=True, mask_distance=mask_distance))
    for i, msa in enumerate(msa_batch_example):
        _check(result[i][0], result[i][1], KAT["mask_distance"][str(mask_distance)][i])
        if mask_distance >= len(msa[0]):
            assert result[i][0] == pytest.approx(gpu_sampler.log_likelihood(msa, target_index=0, with_masking=True)[0], abs=1e-4)


@gpu
@pytest.mark.parametrize("batch_size", [1, 2, 5, 100])
@pytest.mark.parametrize("mask_distance", [1, 2, 5])
def test_likelihood_batch_handles_batch_sizes(gpu_sampler, msa_batch_example, batch_size, mask_distance):
    """:341-354: the forward batch size changes nothing"""
    result = list(gpu_sampler.log_likelihood_batch(msa_batch_example, target_index=0, with_masking=True, mask_distance=mask_distance,
                                                   batch_size=batch_size))
    base = list(gpu_sampler.log_likelihood_batch(msa_batch_example, target_index=0, with_masking=True, mask_distance=mask_distance))
    for i in range(2):
        _check(result[i][0], result[i][1], KAT["mask_distance"][str(mask_distance)][i])
        assert result[i][0] == pytest.approx(base[i][0], abs=1e-4)


# ---- the likelihood_esm_msa front end driven as a function (:356-550) ----------------------------------------------------------------
@gpu
@pytest.mark.parametrize("input_index,mask_off,mask_distance,which", [
    (0, True, float("inf"), "without_mask"), (1, True, float("inf"), "without_mask"),
    (0, False, float("inf"), "with_mask"), (1, False, float("inf"), "with_mask"),
    (0, False, 1, "mask_distance"), (1, False, 1, "mask_distance")])
def test_likelihood_executable_no_mask(gpu_sampler, msa_batch_example, input_index, mask_off, mask_distance, which, tmp_path):
    """:356-398: query = first row, the other rows are the reference alignment; the TSV carries the header and one score, the
    position-wise file one ';'-joined value per residue -- and the score is the sampler's own log_likelihood of that alignment"""
    rows = msa_batch_example[input_index]
    input_handle = StringIO(f">0\n{rows[0]}\n")
    alignment_handle = StringIO("\n".join(f">{n}\n{s}" for n, s in enumerate(rows[1:])) + "\n")
    output_handle = StringIO()
    positionwise = str(tmp_path / "positionwise_output.tsv")
    likelihood_esm_msa.main(input_h=input_handle, output_h=output_handle, masking_off=mask_off, sampler=gpu_sampler,
                            mask_distance=mask_distance, reference_msa_handle=alignment_handle, delete_insertions=False,
                            batch_size=1, subset_strategy="in_order", alignment_size=4, positionwise=positionwise)
    output_handle.seek(0)
    assert output_handle.readline().split() == ["id", "esm-msa"]
    out_n, out_v = output_handle.readline().split()
    assert out_n == "0"
    direct = gpu_sampler.log_likelihood(rows, target_index=0, with_masking=not mask_off, mask_distance=mask_distance)[0]
    assert float(out_v) == pytest.approx(direct, abs=1e-4)
    if PRETRAINED:
        want = KAT["mask_distance"]["1"][input_index] if which == "mask_distance" else KAT[which][input_index]
        assert float(out_v) == pytest.approx(want, abs=2e-3)
    poswise = open(positionwise).readlines()
    assert len(poswise) == 2 and len(poswise[1].split()[1].split(";")) == len(rows[0])


_REALIGN = ["GKEKQ-LDPQYVSQIFHTIIEDSVLYQRS-----", "AKDKG-LDINSAEKFFEALHSESIKHQINVMEK-", "N--EGPLDKESVRTIYELLMSSSHDIQAEQRQRE",
            "GQEQN-LDSNYISQVYHTIIEQSVLSQQEFNNRF", "N--PGPLDDSAIISMFNLIMDGSRILEKKQTNQH"]


def _mafft_realign(msa, new_seq):
    """stands in for `mafft --add` where the binary is absent: puts the gaps of the aligned query (the first row above) back"""
    aligned = {s.replace("-", ""): s for s in _REALIGN}
    return [aligned[new_seq]] + list(msa)


@gpu
def test_likelihood_executable_realign(gpu_sampler, monkeypatch):
    """:401-462: an unaligned query added to the alignment scores like the same query supplied aligned.  `mafft --add` itself is
    used when it is installed; otherwise a stand-in that restores the known alignment of this query (the property under test is the
    front end's handling of `unaligned_queries`, not mafft)"""
    import shutil
    if shutil.which("mafft") is None:
        monkeypatch.setattr(likelihood_esm_msa, "add_to_msa", _mafft_realign, raising=False)
        monkeypatch.setattr(msa_tools, "add_to_msa", _mafft_realign)
    msa_string = "\n".join(f">{n}\n{s}" for n, s in enumerate(_REALIGN[1:])) + "\n"
    scores = []
    for query, unaligned in ((_REALIGN[0], False), (_REALIGN[0].replace("-", ""), True)):
        out = StringIO()
        likelihood_esm_msa.main(input_h=StringIO(f">xyz\n{query}\n"), output_h=out, masking_off=True, sampler=gpu_sampler,
                                reference_msa_handle=StringIO(msa_string), delete_insertions=False, batch_size=1,
                                subset_strategy="in_order", alignment_size=4, unaligned_queries=unaligned)
        out.seek(0)
        assert out.readline().split() == ["id", "esm-msa"]
        name, value = out.readline().split()
        assert name == "xyz"
        scores.append(float(value))
    assert scores[1] == pytest.approx(scores[0], abs=1e-4)


@gpu
def test_log_likelihood_count_gaps(gpu_sampler):
    """:465-478: a row that is nearly all gaps; counting them changes the mean, and the reference's claim about the direction (gaps
    next to gaps are easy) is a statement about the pretrained weights"""
    aln = ["Q-------", "RINVMEK-", "IAEQRQRE", "GQEFNNRF", "FKKQTNQH"]
    no_gaps = gpu_sampler.log_likelihood(aln, target_index=0, with_masking=False, count_gaps=False)
    gaps = gpu_sampler.log_likelihood(aln, target_index=0, with_masking=False, count_gaps=True)
    assert len(no_gaps[1]) == 1 and len(gaps[1]) == 8 and gaps[1][0] == pytest.approx(no_gaps[1][0], abs=1e-6)
    if PRETRAINED:
        assert no_gaps < gaps


@gpu
def test_log_likelihood_count_gaps_2(gpu_sampler):
    """:481-494"""
    aln = ["QQQ-Q-QQ", "RINVMEKH", "IAEQRQRE", "GQEFNNRF", "FKKQTNQH"]
    no_gaps = gpu_sampler.log_likelihood(aln, target_index=0, with_masking=False, count_gaps=False)
    gaps = gpu_sampler.log_likelihood(aln, target_index=0, with_masking=False, count_gaps=True)
    assert len(no_gaps[1]) == 6 and len(gaps[1]) == 8
    assert [gaps[1][i] for i in (0, 1, 2, 4, 6, 7)] == pytest.approx(no_gaps[1], abs=1e-6)
    if PRETRAINED:
        assert no_gaps > gaps


@gpu
def test_likelihood_executable_top_hits(gpu_sampler, monkeypatch):
    """:496-533: `top_hits` picks the context sequences by phmmer score and aligns them with the query; the reference's test only
    requires the call to go through.  phmmer / mafft are used when installed, the suite's deterministic stand-ins otherwise"""
    import shutil
    from _standin import fake_generate_alignment
    if shutil.which("phmmer") is None or shutil.which("mafft") is None:
        for mod in (likelihood_esm_msa, msa_tools):
            monkeypatch.setattr(mod, "run_phmmer", fake_run_phmmer, raising=False)
            monkeypatch.setattr(mod, "generate_alignment", fake_generate_alignment, raising=False)
            monkeypatch.setattr(mod, "add_to_msa", fake_add_to_msa, raising=False)
    seqs = ["GKEKQLDPQYVSQIFHTIIEDSVLYQRS", "AKDKGLDINSAEKFFEALHSESIKHQINVMEK", "NEGPLDKESVRTIYELLMSSSHDIQAEQRQRE",
            "GQEQNLDSNYISQVYHTIIEQSVLSQQEFNNRF", "NPGPLDDSAIISMFNLIMDGSRILEKKQTNQH"]
    out = StringIO()
    likelihood_esm_msa.main(input_h=StringIO(f">xyz\n{seqs[-1]}\n"), output_h=out, masking_off=True, sampler=gpu_sampler,
                            reference_msa_handle=StringIO("\n".join(f">{n}\n{s}" for n, s in enumerate(seqs[:-1])) + "\n"),
                            delete_insertions=False, batch_size=1, subset_strategy="top_hits", alignment_size=2)
    out.seek(0)
    assert out.readline().split() == ["id", "esm-msa"]
    name, value = out.readline().split()
    assert name == "xyz" and float(value) < 0


# ---- partition and generate_single (:536-565) ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("num_partitions,expected", [
    (1, [[1, 2, 3, 4, 5, 6, 7, 8, 9, 10]]), (2, [[1, 2, 3, 4, 5], [6, 7, 8, 9, 10]]), (3, [[1, 2, 3, 4], [5, 6, 7], [8, 9, 10]]),
    (4, [[1, 2, 3], [4, 5, 6], [7, 8], [9, 10]]), (5, [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10]]),
    (6, [[1, 2], [3, 4], [5, 6], [7, 8], [9], [10]]), (7, [[1, 2], [3, 4], [5, 6], [7], [8], [9], [10]]),
    (8, [[1, 2], [3, 4], [5], [6], [7], [8], [9], [10]]), (9, [[1, 2], [3], [4], [5], [6], [7], [8], [9], [10]]),
    (10, [[i] for i in range(1, 11)]), (11, [[i] for i in range(1, 11)]), (600, [[i] for i in range(1, 11)])])
def test_partition_1(num_partitions, expected):
    """:536-556"""
    assert esm_msa_sampler.partition(list(range(1, 11)), num_partitions) == expected


@gpu
def test_generate_single(gpu_sampler):
    """:561-565: one string of the first row's length; with the pretrained weights the reference expects the consensus back"""
    out = gpu_sampler.generate_single(["AAA", "AAA", "GGG"], steps=1, passes=3, burn_in=0)
    assert isinstance(out, str) and len(out) == 3
    assert not set(out) - set(ESM_MSA_ALLOWED_AMINO_ACIDS)
    if PRETRAINED:
        assert out == "AAA"
