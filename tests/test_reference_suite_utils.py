"""The reference's /root/reference/test/test_utils.py and /root/reference/test/test_pgen_msa_revised.py, test by test and under
the same names, against this package (`protein_gibbs_sampler_amd.utils` is the `pgen.utils` surface; the lines each test restates
are in its docstring).  Inputs and expected values are the reference tests' own (data).

mafft and phmmer are not in this image: the three tests that run them use the binaries when present and skip otherwise (the command
lines this package builds are pinned in tests/test_callers_cpu.py::test_tool_wrappers_with_mocked_subprocess); the two
pgen_msa_revised tests run the whole front end on the MI355X with the suite's deterministic stand-ins when the tools are absent."""
import io
import os
import shutil
import warnings

import pytest

from protein_gibbs_sampler_amd import pgen_msa_revised, utils
from _standin import fake_generate_alignment, fake_run_phmmer

PRETRAINED = os.path.exists(os.path.expanduser("~/.cache/torch/hub/checkpoints/esm_msa1b_t12_100M_UR50S.pt"))
needs_mafft = pytest.mark.skipif(shutil.which("mafft") is None, reason="mafft is not installed (the reference's test needs it too)")
needs_phmmer = pytest.mark.skipif(shutil.which("phmmer") is None, reason="phmmer is not installed (the reference's test needs it too)")

_A2M = """
>seq_1
mdgtrtsldieeysdtevqknqvlTLEEWQDKWVNGKTAFHQEQGHQLLKKHLDTflKGKSGLRVFFPLCGKAVEMKWFADRGHSVVGVEISELGIQEFFTEQNLSYSeep*
>seq_2
........................TLEEWQDKWVNGKTAFHQEQGHQLLKKHLDT..KGKSGLRVFFPLCGKAVEMKWFADRGHSVVGVEISELGIQEFFTEQNLSYS...*
>seq_3
mdgtrtsldieeysdtevqknqvlTLEEWQDKWVNGK
TAFHQEQGHQLLKKHLDTflKGKSGLRVFFPLCGKAV
EMKWFADRGHSVVGVEISELGIQEFFTEQNLSYSeep*

"""
_FULL = "mdgtrtsldieeysdtevqknqvlTLEEWQDKWVNGKTAFHQEQGHQLLKKHLDTflKGKSGLRVFFPLCGKAVEMKWFADRGHSVVGVEISELGIQEFFTEQNLSYSeep*"
_DOTS = "........................TLEEWQDKWVNGKTAFHQEQGHQLLKKHLDT..KGKSGLRVFFPLCGKAVEMKWFADRGHSVVGVEISELGIQEFFTEQNLSYS...*"
_MATCH = "TLEEWQDKWVNGKTAFHQEQGHQLLKKHLDTKGKSGLRVFFPLCGKAVEMKWFADRGHSVVGVEISELGIQEFFTEQNLSYS"
_UPPER = "MDGTRTSLDIEEYSDTEVQKNQVLTLEEWQDKWVNGKTAFHQEQGHQLLKKHLDTFLKGKSGLRVFFPLCGKAVEMKWFADRGHSVVGVEISELGIQEFFTEQNLSYSEEP"


@pytest.fixture()
def a2m_file():
    """test_utils.py:7-23: an a2m alignment (lower case / '.' = insert columns, '*' = end marker), one record wrapped"""
    return io.StringIO(_A2M)


# ---- SequenceSubsetter (test_utils.py:27-45) -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("test_input,expected", [
    ({"seq_list": [0, 1, 2, 3, 4, 5], "n": 1, "keep_first": True, "strategy": "in_order"}, [0]),
    ({"seq_list": [0, 1, 2, 3, 4, 5], "n": 1, "keep_first": False, "strategy": "in_order"}, [0]),
    ({"seq_list": [0, 1, 2, 3, 4, 5], "n": 1, "keep_first": True, "strategy": "random"}, [0]),
    ({"seq_list": [0, 1, 2, 3, 4, 5], "n": 3, "keep_first": True, "strategy": "in_order"}, [0, 1, 2]),
    ({"seq_list": [0, 1, 2, 3, 4, 5], "n": 3, "keep_first": False, "strategy": "in_order"}, [0, 1, 2]),
    ({"seq_list": [0, 1, 2, 3, 4, 5], "n": 0, "keep_first": False, "strategy": "in_order"}, []),
    ({"seq_list": [0, 1, 2, 3, 4, 5], "n": 0, "keep_first": True, "strategy": "in_order"}, []),
    ({"seq_list": [0, 1, 2, 3, 4, 5], "n": 5000, "keep_first": True, "strategy": "in_order"}, [0, 1, 2, 3, 4, 5]),
    ({"seq_list": [0, 1, 2, 3, 4, 5], "n": 5000, "keep_first": False, "strategy": "in_order"}, [0, 1, 2, 3, 4, 5]),
])
def test_subsetter_1(test_input, expected):
    """:27-38"""
    assert utils.SequenceSubsetter.subset(**test_input) == expected


def test_subsetter_2():
    """:40-45: a seeded random subset of everything keeps the first sequence first and shuffles the rest"""
    seq_list = [0, 1, 2, 3, 4, 5]
    output = utils.SequenceSubsetter.subset(seq_list=seq_list, n=5000, keep_first=True, strategy="random", random_seed=1)
    assert output[0] == 0 and set(output) == set(seq_list) and output != seq_list


# ---- parse_fasta and its `clean` modes (:47-74) ---------------------------------------------------------------------------------------
def test_parse_fasta_1(a2m_file):
    """:47-52: names and raw sequences, wrapped records joined"""
    names, sequences = utils.parse_fasta(a2m_file, return_names=True)
    assert names == ["seq_1", "seq_2", "seq_3"]
    assert sequences == [_FULL, _DOTS, _FULL]


def test_parse_fasta_2(a2m_file):
    """:54-58: clean="delete" keeps the match columns only"""
    assert utils.parse_fasta(a2m_file, return_names=False, clean="delete") == [_MATCH, _MATCH, _MATCH]


def test_parse_fasta_3_upper(a2m_file):
    """:61-66 (the first of the two functions the reference names test_parse_fasta_3; Python keeps only the second, this suite runs
    both): clean="upper" upper-cases inserts, turns '.' into '-', drops '*'"""
    assert utils.parse_fasta(a2m_file, return_names=False, clean="upper") == [_UPPER, _DOTS.replace(".", "-")[:-1], _UPPER]


def test_parse_fasta_3(a2m_file):
    """:68-73: clean="unalign" additionally drops the gaps"""
    assert utils.parse_fasta(a2m_file, return_names=False, clean="unalign") == [_UPPER, _MATCH, _UPPER]


# ---- gap bookkeeping around generate_single (:76-89) ---------------------------------------------------------------------------------
@pytest.mark.parametrize("test_input,expected", [
    (".*-ABCDE.*-", ("ABCDE", [".", "*", "-", None, None, None, None, None, ".", "*", "-"])),
    ("AB.*-AB", ("ABAB", [None, None, ".", "*", "-", None, None])),
])
def test_unalign_1(test_input, expected):
    """:76-81"""
    assert utils.unalign(test_input) == expected


@pytest.mark.parametrize("test_input,expected", [
    (("ABCDE", [".", "*", "-", None, None, None, None, None, ".", "*", "-"]), ".*-ABCDE.*-"),
    (("ABAB", [None, None, ".", "*", "-", None, None]), "AB.*-AB"),
    (("MTGQ", [None, "-", "-", None, None, ".", "-", None, "*"]), "M--TG.-Q*"),
])
def test_add_gaps_back(test_input, expected):
    """:83-89"""
    assert utils.add_gaps_back(*test_input) == expected


# ---- the external aligner / search tool (:91-141) ------------------------------------------------------------------------------------
@needs_mafft
def test_add_to_msa():
    """:91-108: `mafft --add`: the new sequence comes first, the existing rows keep their order and gain the new gap columns"""
    msa = ["SNNKNQLEHLRTQIDEIDNKLIALIAERLNISRKVGQDKRQLNKQILEKNRYRDLLTHQQRFAKDKG-LDINSAEKFFEALHSESIKHQINVMEK-",
           "-ESQARVAALREKIDELDRRVLVLLSERMAIVQETAAIKRANGGHIYDPKRERALIDRLVAGN--EGPLDKESVRTIYELLMSSSHDIQAEQRQRE",
           "-AVIDALNKTSEQVTEIDNQLINILKERRQLAIAIARAKHQAEKPVRQQDREQQVLARLIQSGQEQN-LDSNYISQVYHTIIEQSVLSQQEFNNRF",
           "-----KVKELRTQVDALDRELLELFNRRASIAMEIGLAKKARGTPVYSPKREKLLLEKMKQTN--PGPLDDSAIISMFNLIMDGSRILEKKQTNQH"]
    new_seq = "-EQAYSLADIRLNVSKLDNDLLDLLSQRRKLAIEVAKAKLKVSKPIRDQEREQELLVKLIETGK-EKQLDPQYVSQIFHTIIEDSVLYQRSFLEQI"
    expected = ["-EQAYSLADIRLNVSKLDNDLLDLLSQRRKLAIEVAKAKLKVSKPIRDQEREQELLVKLIETGK-EKQ-LDPQYVSQIFHTIIEDSVLYQRSFLEQI",
                "SNNKNQLEHLRTQIDEIDNKLIALIAERLNISRKVGQDKRQLNKQILEKNRYRDLLTHQQRFAK-DKG-LDINSAEKFFEALHSESIKHQINVMEK-",
                "-ESQARVAALREKIDELDRRVLVLLSERMAIVQETAAIKRANGGHIYDPKRERALIDRLVAGN---EGPLDKESVRTIYELLMSSSHDIQAEQRQRE",
                "-AVIDALNKTSEQVTEIDNQLINILKERRQLAIAIARAKHQAEKPVRQQDREQQVLARLIQSGQ-EQN-LDSNYISQVYHTIIEQSVLSQQEFNNRF",
                "-----KVKELRTQVDALDRELLELFNRRASIAMEIGLAKKARGTPVYSPKREKLLLEKMKQTN---PGPLDDSAIISMFNLIMDGSRILEKKQTNQH"]
    assert utils.add_to_msa(msa, new_seq) == expected


@needs_mafft
def test_generate_alignment():
    """:110-124: every row named, unique names, equal widths, the input order and residues preserved"""
    seqs = ["AKDKGLDINSAEKFFEALHSESIKHQINVMEK", "NEGPLDKESVRTIYELLMSSSHDIQAEQRQRE", "GQEQNLDSNYISQVYHTIIEQSVLSQQEFNNRF",
            "NPGPLDDSAIISMFNLIMDGSRILEKKQTNQH", "GKEKQLDPQYVSQIFHTIIEDSVLYQRS"]
    names, rows = utils.generate_alignment({"1": seqs})
    assert len(names) == len(rows) == len(seqs) and len(set(names)) == len(names)
    assert len({len(r) for r in rows}) == 1
    assert [r.replace("-", "") for r in rows] == seqs


@needs_phmmer
def test_run_phmmer(tmp_path):
    """:126-141: hits come back as database record names, best first"""
    db = ["MTFKLPDLPFDAGALEPYISALTMKTHHGKHHAAYIKNMNAILAERADAQTSLEAVVSLAAREANKKLFNNAAQAWNHGFFWQSLSADAQNGPSGDLRAAIMNSFGSLEAFNDEAKAKGVGHFASGWLWLVSDESGALSLCDLHDADTPITDPSLTPLLVCDLWEHAYYIDYANERPRFVDAFLTKLANWRFAQAQYQAARSGSGA",
          "FAVSATKIHTKATLPALDYAYEALEPILSSHLLHLHHDKHHQTYVNNLNAAEEKLKDPSLDLHTQIALQSAIKFNGGGHVNHSIYWKNLAPKSAGGGAFNAQAPLGQAIVKKWGSFEAFKKNFNTQLAAIQGSGWGWLIKDADGSLRITTTMNQDTILDATPVITIDAWEHAYYPQYENRKAEYYENIWQIINWKEAEAR",
          "MKFELPALPYPVNALEPTMSARTIEFHWGKHEAAYINNLNGLIEGTPLENDTLEEIVRKSDGPIYNNAAQAWNHIFFFFQLAPNGKKEPGGALAEAIDRHFGSFAAFKEAFAKAGATLFGSGWAWLSVKPDGQLEITQGPNAHNPLKNGAVPLLTADVWEHAYYLDYQNRRPDFLSALWNLVDWKVIEKR",
          "MTHALPELGYDYDALEPFIDAKTMEIHHTKHHQTYVDKLNAALDGHDDLAKLGVNELISDLGKVPESIRPAVRNHGGGHSNHSFFWPLLKKNVALGGAVQEAIDRDFGSFDSFKTEFSNKAALLFGSGWTWVVADQGKLSIVTTPNQDSPVSDGKTPVLGLDVWEHAYYLKYQNRRPDYINAFFDIINWDKVNG"]
    query = "MSFELPALPYAKDALAPHISAETIEYHYGKHHQTYVTNLNNLIKGTAFEGKSLEEIIRSSEGGVFNNAAQVWNHTFYWNCLAPNAGGEPTGKVAEAIAASFGSFADFKAQFTDAAIKNFGSGWTWLVKNSDGKLAIVSTSNAGTPLTTDATPLLTVDVWEHAYYIDYRNARPGYLEHFWALVNWEFVAKNL"
    db_file = str(tmp_path / "tmp.fasta")
    utils.write_sequential_fasta(db_file, db)
    assert utils.run_phmmer(query, db_file) == ["2", "0", "3", "1"]


# ---- pgen_msa_revised end to end (test_pgen_msa_revised.py:10-35) ---------------------------------------------------------------------
_TEMPLATES = ">query_seq1\nMAGIC\n>query_seq2\nMEADAL\n"                                       # test/data/test_query_seqs.fasta
_REFERENCES = ">test_seq1\nMAGIC\n>test_seq2\nMGIC\n>test_seq3\nMLGIC\n>test_seq4\nMEDAL\n>test_seq5\nMEADGL\n"   # test_reference_seqs.fasta
_REFERENCES_SMALL = ">test_seq1\nGLC\n>test_seq2\nGLC\n>test_seq3\nGLC\n>test_seq4\nEADI\n>test_seq5\nEADI\n"      # test_reference_seqs_small.fasta


def _run_front_end(tmp_path, monkeypatch, references, extra):
    if shutil.which("phmmer") is None or shutil.which("mafft") is None:
        monkeypatch.setattr(pgen_msa_revised, "run_phmmer", fake_run_phmmer)
        monkeypatch.setattr(pgen_msa_revised, "generate_alignment", fake_generate_alignment)
    t, r, o = tmp_path / "test_query_seqs.fasta", tmp_path / "refs.fasta", tmp_path / "generated.fasta"
    t.write_text(_TEMPLATES)
    r.write_text(references)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pgen_msa_revised.main(["--templates", str(t), "--references", str(r), "-o", str(o), "--seqs_per_template", "2"]
                              + ([] if PRETRAINED else ["--synthetic-weights"]) + extra)
    names, seqs = utils.parse_fasta(str(o), return_names=True)
    assert names == ["0_query_seq1", "1_query_seq1", "0_query_seq2", "1_query_seq2"]
    # the front end strips the gaps a draw may produce ('-' is a token the MSA sampler may emit, pgen_msa_revised.py:112): the
    # pretrained model leaves none in these alignments (the reference's expectation: 5, 5, 6, 6), seeded synthetic weights may
    assert all(not set(s) - set("ACDEFGHIKLMNPQRSTVWY") for s in seqs)
    if PRETRAINED:
        assert [len(s) for s in seqs] == [5, 5, 6, 6]
    else:
        assert all(len(s) <= n for s, n in zip(seqs, [5, 5, 6, 6]))


@pytest.mark.gpu
def test_pgen_msa_revised_legacy_1(tmp_path, monkeypatch):
    """:10-21: --legacy (the template goes last, the last row is resampled), alignments of one sequence; two new sequences per
    template, named <i>_<template>, as long as their templates.  (--synthetic-weights where there is no checkpoint.)"""
    _run_front_end(tmp_path, monkeypatch, _REFERENCES, ["--alignment_size", "1", "--legacy"])


@pytest.mark.gpu
def test_pgen_msa_revised_1(tmp_path, monkeypatch):
    """:23-35: the current mode with --debug (phmmer --max) and a gap threshold"""
    _run_front_end(tmp_path, monkeypatch, _REFERENCES_SMALL, ["--alignment_size", "5", "--debug", "--gap_percent_threshold", "49"])
