"""The caller modules end to end on the GPU, against text the reference's own modules wrote when they were driven with
the same stand-in model and the same fake phmmer / mafft / muscle (tests/golden/callers.json): pgen_msa_revised pipeline,
pgen_esm_from_fasta, likelihood_esm, likelihood_esm_msa.  With top_k=1 / burn_in=0 all draws are argmaxes, so generated
FASTA must match character for character; log-likelihood tables match to 5e-6 (fp32 log-softmax on the GPU vs torch CPU)."""
import io
import random
import types
import warnings
from pathlib import Path

import pytest

from protein_gibbs_sampler_amd import (esm_msa_sampler, esm_sampler, likelihood_esm, likelihood_esm_msa, pgen_esm_from_fasta,
                                       pgen_msa_revised)
from protein_gibbs_sampler_amd.alphabet import Alphabet
from _standin import fake_add_to_msa, fake_generate_alignment, fake_run_phmmer, load_json, make_standin_torch_module

pytestmark = pytest.mark.gpu
G = load_json("callers.json")
IN = G["inputs"]


class _Plugin:
    def __init__(self, msa, context=False):
        self.alphabet = Alphabet(True, not msa)
        self.batch_converter = self.alphabet.get_batch_converter(msa=msa)
        self.model = make_standin_torch_module(context=context)


def _tables_close(got, want, tol=5e-6):
    gl, wl = got.strip().split("\n"), want.strip().split("\n")
    assert len(gl) == len(wl) and gl[0] == wl[0]
    for a, b in zip(gl[1:], wl[1:]):
        sep = "," if "," in b else "\t"
        (na, va), (nb, vb) = a.split(sep), b.split(sep)
        assert na == nb
        fa, fb = [float(x) for x in va.split(";")], [float(x) for x in vb.split(";")]
        assert len(fa) == len(fb)
        # positionwise values are rounded to 3 decimals in the file: allow one unit in the last place
        assert all(abs(x - y) <= (1.001e-3 if ";" in vb else tol) for x, y in zip(fa, fb)), (a, b)


@pytest.mark.parametrize("idx", range(len(G["pgen_msa_revised"])))
def test_pgen_msa_revised_pipeline(idx, tmp_path, monkeypatch):
    c = G["pgen_msa_revised"][idx]
    kw = c["kw"]
    monkeypatch.setattr(pgen_msa_revised, "run_phmmer", fake_run_phmmer)
    monkeypatch.setattr(pgen_msa_revised, "generate_alignment", fake_generate_alignment)
    t, r, o = tmp_path / "t.fasta", tmp_path / "r.fasta", tmp_path / "o.fasta"
    t.write_text(IN["templates"])
    r.write_text(IN["references"])
    s = esm_msa_sampler.ESM_MSA_sampler(_Plugin(True, context=True), device="cuda:0")
    s.draw_seed = 0
    random.seed(c["pyseed"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pgen_msa_revised.pgen_msa(str(t), str(r), str(o), kw["seqs_per_template"], kw["keep_identical"], kw["steps"], kw["passes"],
                                  kw["burn_in"], "cuda:0", "esm_msa1", kw["alignment_size"], 0.0, 1.53, kw["top_k"], legacy=kw["legacy"],
                                  gap_percent_threshold=kw["gap_percent_threshold"], debug=False, sampler=s)
    assert o.read_text() == c["output"]
    assert random.getrandbits(32) == c["next32"]          # the position shuffles consumed the same Python RNG stream


def test_pgen_msa_revised_warns_on_few_hits(tmp_path, monkeypatch):
    monkeypatch.setattr(pgen_msa_revised, "run_phmmer", lambda *a, **k: [])
    monkeypatch.setattr(pgen_msa_revised, "generate_alignment", fake_generate_alignment)
    t, r, o = tmp_path / "t.fasta", tmp_path / "r.fasta", tmp_path / "o.fasta"
    t.write_text(">q\nMAGIC\n")
    r.write_text(IN["references"])
    s = esm_msa_sampler.ESM_MSA_sampler(_Plugin(True), device="cuda:0")
    with pytest.warns(UserWarning, match="fewer than 4 hits found for template seq q"):
        pgen_msa_revised.pgen_msa(str(t), str(r), str(o), 2, False, 10, 1, 1, "cuda:0", "esm_msa1", 5, 0.0, 1.53, 1, legacy=True, sampler=s)
    names = [line[1:] for line in o.read_text().split("\n") if line.startswith(">")]
    assert names == ["0_q", "1_q"]                         # the reference's own test checks names and lengths (test_pgen_msa_revised.py:10-23)
    seqs = [x for x in o.read_text().split("\n")[1::2] if x]
    # 5 residues unless a gap was drawn (the pipeline strips '-' from its output, pgen_msa_revised.py:113)
    assert len(seqs) == 2 and all(len(x) <= 5 and set(x) <= set("ACDEFGHIKLMNPQRSTVWY") for x in seqs)


@pytest.mark.parametrize("idx", range(len(G["pgen_esm_from_fasta"])))
def test_pgen_esm_from_fasta(idx, tmp_path, capsys):
    c = G["pgen_esm_from_fasta"][idx]
    fa = tmp_path / "seeds.fasta"
    fa.write_text(IN["fasta_gapped"])
    args = types.SimpleNamespace(model="esm1b", device="cuda:0", num_output_sequences=3, batch_size=1, keep_gap_positions=c["keep_gap_positions"])
    s = esm_sampler.ESM_sampler(_Plugin(False), device="cuda:0")
    s.draw_seed = 0
    random.seed(c["pyseed"])
    pgen_esm_from_fasta.main(io.StringIO(c["spec"].replace("{FASTA}", str(fa))), Path(tmp_path), args, sampler=s)
    for name, want in c["files"].items():
        assert (tmp_path / name).read_text() == want.replace("{FASTA}", str(fa)), name
    assert random.getrandbits(32) == c["next32"]


@pytest.mark.parametrize("idx", range(len(G["likelihood_esm"])))
def test_likelihood_esm_tables(idx, tmp_path):
    c = G["likelihood_esm"][idx]
    kw = c["kw"]
    pos = str(tmp_path / "pos.tsv") if kw["positionwise"] else None
    buf = io.StringIO()
    s = esm_sampler.ESM_sampler(_Plugin(False), device="cuda:0")
    likelihood_esm.main(io.StringIO(IN["fasta_gapped"] + IN["queries"]), buf, kw["masking_off"], "cuda:0", "esm1v", kw["batch_size"],
                        float("inf") if kw["mask_distance"] is None else kw["mask_distance"], kw["csv"], kw["score_name"], pos, sampler=s)
    _tables_close(buf.getvalue(), c["table"])
    if pos:
        _tables_close(open(pos).read(), c["positionwise"])


@pytest.mark.parametrize("idx", range(len(G["likelihood_esm_msa"])))
def test_likelihood_esm_msa_tables(idx, tmp_path, monkeypatch):
    c = G["likelihood_esm_msa"][idx]
    kw = c["kw"]
    monkeypatch.setattr(likelihood_esm_msa, "run_phmmer", fake_run_phmmer)
    monkeypatch.setattr(likelihood_esm_msa, "generate_alignment", fake_generate_alignment)
    monkeypatch.setattr(likelihood_esm_msa, "add_to_msa", fake_add_to_msa)
    in_msas = {"q1": ["ACDEFGHIKL", "AC-EFGHIKL", "MCDEFGHIKV"], "q2": ["AC-EFGHIKL", "ACDEFGHIKL"],
               "q3": ["MCDEFGHIKV", "ACDEFG--KL", "ACDEFGHIKL", "A-DEFGHIKL"]}
    pos = str(tmp_path / "pos.tsv") if kw.get("positionwise") else None
    buf = io.StringIO()
    s = esm_msa_sampler.ESM_MSA_sampler(_Plugin(True, context=True), device="cuda:0")
    unal = kw.get("unaligned_reference")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        likelihood_esm_msa.main(input_h=io.StringIO(IN["templates"] if unal else IN["queries"]), output_h=buf,
                                masking_off=kw.get("masking_off", False), sampler=s,
                                reference_msa_handle=io.StringIO(IN["references"] if unal else IN["ref_msa"]),
                                in_msas=in_msas if kw.get("in_msas") else None, batch_size=kw["batch_size"],
                                subset_strategy=kw.get("subset_strategy", "random"), alignment_size=kw.get("alignment_size", 2 ** 63 - 1),
                                subset_random_seed=kw.get("subset_random_seed"), unaligned_queries=kw.get("unaligned_queries", False),
                                mask_distance=kw.get("mask_distance", float("inf")), csv=kw.get("csv", False), positionwise=pos)
    _tables_close(buf.getvalue(), c["table"])
    if pos:
        _tables_close(open(pos).read(), c["positionwise"])
