#!/usr/bin/env python3
"""Golden fixtures for the callers either side of the Gibbs path: the reference's
`pgen_msa_revised.pgen_msa`, `pgen_esm_from_fasta.main`, `likelihood_esm.main`, `likelihood_esm_msa.main` and the pure
helpers of `utils.py` / `pgen_msa_revised.py`, driven HERE (build container only) with

  * the deterministic stand-in model of make_golden.py in place of the fair-esm checkpoints,
  * tests/_standin.py's fake phmmer / mafft / muscle in place of the external programs (absent from the image),
  * empty stand-in modules for `esm` and `Bio` so that the reference modules import.

Nothing of the reference is copied: its modules are imported from /root/reference/src and the fixture holds the inputs
and the text they wrote.   python tests/golden/make_golden_callers.py  ->  tests/golden/callers.json
"""
import io
import json
import os
import random
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference/src")
sys.dont_write_bytecode = True

for name in ("esm", "Bio"):                      # import-time dependencies the image lacks; never called
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules["esm"].pretrained = types.SimpleNamespace()
sys.modules["esm"].data = types.SimpleNamespace(BatchConverter=object, Alphabet=object)
sys.modules["Bio"].SearchIO = types.SimpleNamespace()

import torch  # noqa: E402

from _standin import fake_add_to_msa, fake_generate_alignment, fake_run_phmmer  # noqa: E402
from make_golden import _standin  # noqa: E402

TEMPLATES = ">query_seq1 first template\nMAGICKLV\n>query_seq2\nMEADALQRST\n"
REFERENCES = "".join(f">r{i}\n{s}\n" for i, s in enumerate(
    ["MAGICKLV", "MAGLCKIV", "GICKLVAA", "EADALQRS", "MEADAIQRST", "WWWWWW", "ADALQ", "MAGICKLVMEADAL", "CKLVQRST", "MEAGICST"]))
FASTA_GAPPED = ">s1 desc\nMK-V.AA*\n>s2\nac-DEF\n>s3\n--GHIK--\n"
QUERIES = ">q1\nACDEFGHIKL\n>q2 second\nAC-EFGHIKL\n>q3\nMCDEFGHIKV\n"
REF_MSA = "".join(f">m{i}\n{s}\n" for i, s in enumerate(
    ["ACDEFGHIKL", "AC-EFGHIKL", "ACDEFG--KL", "MCDEFGHIKV", "ACDEYGHIKL", "A-DEFGHIKL", "ACDEFGHIK-"]))


def _write(tmp, name, text):
    p = os.path.join(tmp, name)
    with open(p, "w") as f:
        f.write(text)
    return p


def gen():
    from pgen import utils, pgen_msa_revised, pgen_esm_from_fasta, likelihood_esm, likelihood_esm_msa
    from pgen import esm_msa_sampler as refm
    from pgen import esm_sampler as ref
    out = {"inputs": dict(templates=TEMPLATES, references=REFERENCES, fasta_gapped=FASTA_GAPPED, queries=QUERIES, ref_msa=REF_MSA)}

    # ---- pure helpers
    msas = [["AC-E", "A--E", "-CDE"], ["MK", "M-"], ["----", "A-C-", "A---", "AB--", "----"]]
    out["helpers"] = dict(
        unalign=[[s, list(utils.unalign(s))] for s in (".*-ABCDE.*-", "AB.*-AB", "ac-d1E", "", "---")],
        add_gaps_back=[[a, m, utils.add_gaps_back(a, m)] for a, m in (("ABCDE", [".", "*", "-", None, None, None, None, None, ".", "*", "-"]),
                                                                      ("MTGQ", [None, "-", "-", None, None, ".", "-", None, "*"]), ("", ["-"]))],
        delete_msa_cols=[[m, c, pgen_msa_revised.delete_msa_cols(m, c)] for m in msas for c in ([], [0], [1, 3], [0, 1, 2, 3, 7])],
        count_gaps=[[m, pgen_msa_revised.count_gaps_per_column(m)] for m in msas],
        gap_threshold=[[m, t, pgen_msa_revised.apply_gap_threshold(m, t)] for m in msas for t in (0, 33, 34, 49, 50, 80, 100)],
    )

    # ---- pgen_msa_revised pipeline (fake phmmer/mafft, stand-in model; top_k=1 burn_in=0 => deterministic strings)
    pgen_msa_revised.run_phmmer = fake_run_phmmer
    pgen_msa_revised.generate_alignment = fake_generate_alignment
    pgen_msa_revised.model_map = {"esm_msa1": lambda: _standin("msa1b", context=True)}
    pgen_msa_revised.tqdm = lambda *a, **k: types.SimpleNamespace(__enter__=lambda s: s, __exit__=lambda s, *e: False, update=lambda n: None)

    class _Bar:
        def __init__(self, *a, **k): pass
        def __enter__(self): return self
        def __exit__(self, *e): return False
        def update(self, n): pass
    pgen_msa_revised.tqdm = _Bar
    cases = []
    for kw in (dict(alignment_size=4, seqs_per_template=2, steps=3, passes=2, burn_in=0, top_k=1, legacy=False, gap_percent_threshold=49, keep_identical=False),
               dict(alignment_size=4, seqs_per_template=1, steps=2, passes=1, burn_in=0, top_k=1, legacy=True, gap_percent_threshold=80, keep_identical=False),
               dict(alignment_size=3, seqs_per_template=1, steps=4, passes=2, burn_in=0, top_k=1, legacy=False, gap_percent_threshold=20, keep_identical=True),
               dict(alignment_size=1, seqs_per_template=2, steps=10, passes=1, burn_in=0, top_k=1, legacy=True, gap_percent_threshold=80, keep_identical=False)):
        with tempfile.TemporaryDirectory() as tmp:
            t, r, o = _write(tmp, "t.fasta", TEMPLATES), _write(tmp, "r.fasta", REFERENCES), os.path.join(tmp, "o.fasta")
            random.seed(5)
            torch.manual_seed(5)
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                pgen_msa_revised.pgen_msa(t, r, o, kw["seqs_per_template"], kw["keep_identical"], kw["steps"], kw["passes"], kw["burn_in"], "cpu",
                                          "esm_msa1", kw["alignment_size"], 0.0, 1.53, kw["top_k"], legacy=kw["legacy"],
                                          gap_percent_threshold=kw["gap_percent_threshold"], debug=False)
            cases.append(dict(kw=kw, pyseed=5, output=open(o).read(), next32=random.getrandbits(32)))
    out["pgen_msa_revised"] = cases

    # ---- pgen_esm_from_fasta
    pgen_esm_from_fasta.model_map = {"esm1b": lambda: _standin("esm1b")}
    pgen_esm_from_fasta.trange = range
    cases = []
    for keep in (False, True):
        with tempfile.TemporaryDirectory() as tmp:
            fa = _write(tmp, "seeds.fasta", FASTA_GAPPED)
            spec = "job1\t{'num_iters': 3, 'burnin': 0, 'top_k': 1, 'num_positions': 2}\t%s\n\nbad line\njob2\t{'num_iters': 2, 'burnin': 0, 'top_k': 1, 'in_order': True, 'num_positions': 1}\t%s\n" % (fa, fa)
            args = types.SimpleNamespace(model="esm1b", device="cpu", num_output_sequences=3, batch_size=1, keep_gap_positions=keep)
            from pathlib import Path
            random.seed(9)
            torch.manual_seed(9)
            stdout, sys.stdout = sys.stdout, io.StringIO()
            try:
                pgen_esm_from_fasta.main(io.StringIO(spec), Path(tmp), args)
            finally:
                sys.stdout = stdout
            cases.append(dict(keep_gap_positions=keep, pyseed=9, spec=spec.replace(fa, "{FASTA}"),
                              files={n: open(os.path.join(tmp, n)).read().replace(fa, "{FASTA}") for n in ("specification.tsv", "job1.fasta", "job2.fasta")},
                              next32=random.getrandbits(32)))
    out["pgen_esm_from_fasta"] = cases

    # ---- likelihood_esm
    likelihood_esm.model_map = {"esm1v": lambda: _standin("esm1b")}
    likelihood_esm.tqdm = types.SimpleNamespace(trange=range)
    cases = []
    for kw in (dict(masking_off=False, batch_size=1, mask_distance=float("inf"), csv=False, score_name=None, positionwise=True),
               dict(masking_off=False, batch_size=2, mask_distance=3, csv=True, score_name="myscore", positionwise=True),
               dict(masking_off=True, batch_size=3, mask_distance=float("inf"), csv=False, score_name=None, positionwise=False)):
        with tempfile.TemporaryDirectory() as tmp:
            pos = os.path.join(tmp, "pos.tsv") if kw["positionwise"] else None
            buf = io.StringIO()
            likelihood_esm.main(io.StringIO(FASTA_GAPPED + QUERIES), buf, kw["masking_off"], "cpu", "esm1v", kw["batch_size"], kw["mask_distance"],
                                kw["csv"], kw["score_name"], pos)
            cases.append(dict(kw={k: (None if v == float("inf") else v) for k, v in kw.items()}, table=buf.getvalue(),
                              positionwise=open(pos).read() if pos else None))
    out["likelihood_esm"] = cases

    # ---- likelihood_esm_msa
    likelihood_esm_msa.run_phmmer = fake_run_phmmer
    likelihood_esm_msa.generate_alignment = fake_generate_alignment
    likelihood_esm_msa.add_to_msa = fake_add_to_msa
    likelihood_esm_msa.tqdm = types.SimpleNamespace(tqdm=lambda it, **k: it)
    cases = []
    in_msas = {"q1": ["ACDEFGHIKL", "AC-EFGHIKL", "MCDEFGHIKV"], "q2": ["AC-EFGHIKL", "ACDEFGHIKL"], "q3": ["MCDEFGHIKV", "ACDEFG--KL", "ACDEFGHIKL", "A-DEFGHIKL"]}
    for kw in (dict(subset_strategy="in_order", alignment_size=3, batch_size=1, positionwise=True),
               dict(subset_strategy="random", alignment_size=4, subset_random_seed=7, batch_size=1, csv=True, positionwise=True, mask_distance=4),
               dict(subset_strategy="in_order", alignment_size=5, unaligned_queries=True, batch_size=1, masking_off=True),
               dict(subset_strategy="top_hits", alignment_size=3, batch_size=1, positionwise=True, mask_distance=2, unaligned_reference=True),
               dict(in_msas=True, batch_size=1, positionwise=True)):
        with tempfile.TemporaryDirectory() as tmp:
            pos = os.path.join(tmp, "pos.tsv") if kw.get("positionwise") else None
            buf = io.StringIO()
            sampler = refm.ESM_MSA_sampler(_standin("msa1b", context=True), device="cpu")
            ref_text = REFERENCES if kw.get("unaligned_reference") else REF_MSA
            q_text = TEMPLATES if kw.get("unaligned_reference") else QUERIES
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                likelihood_esm_msa.main(input_h=io.StringIO(q_text), output_h=buf, masking_off=kw.get("masking_off", False), sampler=sampler,
                                        reference_msa_handle=io.StringIO(ref_text), in_msas=in_msas if kw.get("in_msas") else None,
                                        batch_size=kw["batch_size"], subset_strategy=kw.get("subset_strategy", "random"),
                                        alignment_size=kw.get("alignment_size", sys.maxsize), subset_random_seed=kw.get("subset_random_seed"),
                                        unaligned_queries=kw.get("unaligned_queries", False), mask_distance=kw.get("mask_distance", float("inf")),
                                        csv=kw.get("csv", False), positionwise=pos)
            cases.append(dict(kw=kw, table=buf.getvalue(), positionwise=open(pos).read() if pos else None))
    out["likelihood_esm_msa"] = cases

    with open(os.path.join(HERE, "callers.json"), "w") as f:
        json.dump(out, f)
    print("callers.json written:", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    gen()
