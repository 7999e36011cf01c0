"""Host-side callers either side of the Gibbs path: alignment helpers against outputs recorded
from the reference (tests/golden/callers.json, made by tests/golden/make_golden_callers.py), the reference's own KATs
for `unalign` / `add_gaps_back` (test/test_utils.py:76-89), the HMMER3 report parser and the command-line surfaces."""
import subprocess

import pytest

from protein_gibbs_sampler_amd import (clean_fasta, likelihood_esm, likelihood_esm_msa, msa_tools, pgen_esm_from_fasta,
                                       pgen_msa_revised)
from _standin import load_json

G = load_json("callers.json")


def test_helpers_match_reference_recordings():
    h = G["helpers"]
    for s, (clean, mask) in h["unalign"]:
        assert list(msa_tools.unalign(s)) == [clean, mask]
    for a, m, want in h["add_gaps_back"]:
        assert msa_tools.add_gaps_back(a, m) == want
    for m, c, want in h["delete_msa_cols"]:
        assert msa_tools.delete_msa_cols(m, c) == want
    for m, want in h["count_gaps"]:
        assert msa_tools.count_gaps_per_column(m) == want
    for m, t, want in h["gap_threshold"]:
        assert msa_tools.apply_gap_threshold(m, t) == want


@pytest.mark.parametrize("seq,expected", [
    (".*-ABCDE.*-", ("ABCDE", [".", "*", "-", None, None, None, None, None, ".", "*", "-"])),
    ("AB.*-AB", ("ABAB", [None, None, ".", "*", "-", None, None])),
])
def test_unalign_kat(seq, expected):
    assert msa_tools.unalign(seq) == expected
    assert msa_tools.add_gaps_back(*expected) == seq


def test_add_gaps_back_kat():
    assert msa_tools.add_gaps_back("MTGQ", [None, "-", "-", None, None, ".", "-", None, "*"]) == "M--TG.-Q*"


def test_write_partitioned_fasta(tmp_path):
    p = tmp_path / "x.fasta"
    msa_tools.write_partitioned_fasta(p, {"1": ["AAA", "CC"], "ref": ["G"]})
    assert p.read_text() == ">1_0\nAAA\n>1_1\nCC\n>ref_0\nG\n"


PHMMER_REPORT = """# phmmer :: search a protein sequence against a protein database
# HMMER 3.3.2 (Nov 2020); http://hmmer.org/
# - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - -
# query sequence file:             /tmp/x/query.fa
# target sequence database:        /tmp/db.fasta
# - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - -

Query:       QUERY  [L=193]
Scores for complete sequences (score includes all domains):
   --- full sequence ---   --- best 1 domain ---    -#dom-
    E-value  score  bias    E-value  score  bias    exp  N  Sequence Description
    ------- ------ -----    ------- ------ -----   ---- --  -------- -----------
    1.1e-58  187.1   0.1    1.3e-58  186.9   0.1    1.0  1  2         
    2.2e-40  127.5   0.0    2.5e-40  127.3   0.0    1.0  1  0         some description here
  ------ inclusion threshold ------
      0.012   12.0   0.0      0.015   11.7   0.0    1.2  1  3         


Domain annotation for each sequence:
>> 2  
   #    score  bias  c-Evalue  i-Evalue hmmfrom  hmm to    alifrom  ali to    envfrom  env to     acc
 ---   ------ ----- --------- --------- ------- -------    ------- -------    ------- -------    ----
   1 !  186.9   0.1   9.9e-59   1.3e-58       2     192 ..       1     190 [.       1     191 [. 0.98

Internal pipeline statistics summary:
-------------------------------------
Query sequence(s):                         1  (193 residues searched)
//
[ok]
"""


def test_parse_phmmer_hits():
    assert msa_tools.parse_phmmer_hits(PHMMER_REPORT) == ["2", "0", "3"]
    empty = PHMMER_REPORT.split("    1.1e-58")[0] + "\n   [No hits detected that satisfy reporting thresholds]\n\n\nDomain annotation for each sequence:\n"
    assert msa_tools.parse_phmmer_hits(empty) == []


def test_external_tools_fail_loudly_when_absent(monkeypatch):
    def missing(*a, **k):
        raise FileNotFoundError
    monkeypatch.setattr(subprocess, "run", missing)
    for call in (lambda: msa_tools.run_phmmer("ACD", "/nonexistent.fasta"), lambda: msa_tools.generate_alignment({"1": ["ACD", "AC"]}),
                 lambda: msa_tools.add_to_msa(["ACD"], "AC")):
        with pytest.raises(Exception, match="not found on PATH"):
            call()


def test_command_line_surfaces():
    """Every flag of the reference front ends is accepted (pgen_msa_revised.py:119-146, pgen_esm_from_fasta.py:62-68,
    likelihood_esm.py:65-74, likelihood_esm_msa.py:154-175, clean_fasta.py:7-11)."""
    a = pgen_msa_revised.build_parser().parse_args(
        "--templates t.fa --references r.fa -o out.fa --seqs_per_template 2 --keep_identical --steps 5 --passes 2 --burn_in 1 "
        "--top_k 3 --legacy --gap_percent_threshold 49 --ep 0.1 --op 2 --device cuda:0 --model esm_msa1 --alignment_size 8 --debug".split())
    assert (a.seqs_per_template, a.steps, a.passes, a.burn_in, a.top_k, a.alignment_size) == (2, 5, 2, 1, 3, 8)
    assert a.legacy and a.debug and a.keep_identical and a.gap_percent_threshold == 49.0 and (a.ep, a.op) == (0.1, 2.0)
    d = pgen_msa_revised.build_parser().parse_args("--templates t --references r -o o".split())
    assert (d.steps, d.passes, d.burn_in, d.top_k, d.alignment_size, d.gap_percent_threshold, d.ep, d.op) == (10, 3, 1, 1, 32, 80.0, 0.0, 1.53)
    b = pgen_esm_from_fasta.build_parser().parse_args("-o out -i spec.tsv --num_output_sequences 4 --model esm1b --keep_gap_positions".split())
    assert b.num_output_sequences == 4 and b.keep_gap_positions and b.batch_size == 1
    with pytest.raises(SystemExit):
        pgen_esm_from_fasta.build_parser().parse_args("--batch_size 2".split())
    c = likelihood_esm.build_parser().parse_args("-i in.fa -o out.tsv --batch_size 4 --masking_off --model esm1b --csv --score_name s "
                                                 "--positionwise p.tsv".split())
    assert c.batch_size == 4 and c.masking_off and c.csv and c.score_name == "s" and c.positionwise == "p.tsv" and c.mask_distance is None
    assert likelihood_esm.build_parser().parse_args([]).model == "esm1v"
    e = likelihood_esm_msa.build_parser().parse_args(
        "-i in.fa -o o.tsv --reference_msa ref.fa --delete_insertions --alignment_size 31 --keep_identical --batch_size 2 "
        "--subset_strategy top_hits --subset_random_seed 3 --redraw --unaligned_queries --mask_distance 5 --csv --positionwise p".split())
    assert e.subset_strategy == "top_hits" and e.alignment_size == 31 and e.subset_random_seed == 3 and e.redraw and e.unaligned_queries
    with pytest.raises(SystemExit):
        likelihood_esm_msa.build_parser().parse_args([])          # --reference_msa is required


def test_likelihood_cli_argument_errors(monkeypatch):
    with pytest.raises(ValueError, match="mask distance must be an integer >= 1"):
        likelihood_esm.cli(["--mask_distance", "0"])
    with pytest.raises(ValueError, match="both set"):
        likelihood_esm.cli(["--mask_distance", "2", "--masking_off"])
    with pytest.raises(ValueError, match="redraw is set"):
        likelihood_esm_msa.cli(["--reference_msa", "x", "--redraw", "--subset_strategy", "in_order"])


def test_clean_fasta(tmp_path):
    src, dst = tmp_path / "in.fa", tmp_path / "out.fa"
    src.write_text(G["inputs"]["fasta_gapped"])
    clean_fasta.main(["-i", str(src), "-o", str(dst), "--clean_strategy", "unalign"])
    assert dst.read_text() == ">s1\nMKVAA\n>s2\nACDEF\n>s3\nGHIK\n"
    clean_fasta.main(["-i", str(src), "-o", str(dst), "--clean_strategy", "delete", "--full_name"])
    assert dst.read_text() == ">s1 desc\nMK-VAA\n>s2\n-DEF\n>s3\n--GHIK--\n"


def test_tool_wrappers_with_mocked_subprocess(monkeypatch, tmp_path):
    """run_phmmer / generate_alignment / add_to_msa: the command lines the reference uses (utils.py:251,294-297,196) and the
    parsing of what comes back, with subprocess.run replaced by canned phmmer / mafft / muscle outputs."""
    calls = []

    class Res:
        def __init__(self, out, rc=0, as_bytes=False):
            self.stdout = out.encode() if as_bytes else out
            self.stderr = b"" if as_bytes else ""
            self.returncode = rc

    def fake_run(argv, **kw):
        calls.append(list(argv))
        if argv[0] == "phmmer":
            assert open(argv[-2]).read() == ">QUERY\nMKVA\n"            # the query FASTA the wrapper wrote
            return Res(PHMMER_REPORT)
        if argv[0] == "mafft":
            seqs = [l.strip() for l in open(argv[-1]) if not l.startswith(">")]
            names = [l.strip()[1:] for l in open(argv[-1]) if l.startswith(">")]
            width = max(map(len, seqs))
            return Res("".join(">%s\n%s\n" % (n, s + "-" * (width - len(s))) for n, s in zip(names, seqs)), as_bytes=True)
        if argv[0] == "muscle":
            old = [l.strip() for l in open(argv[argv.index("-in1") + 1]) if not l.startswith(">")]
            return Res("".join(">%d\n%s-\n" % (i, s) for i, s in enumerate(old)) + ">new_seq\nMKVAA\n")
        raise AssertionError(argv)

    monkeypatch.setattr(subprocess, "run", fake_run)
    assert msa_tools.run_phmmer("MKVA", tmp_path / "db.fasta", max_mode=True) == ["2", "0", "3"]
    assert calls[-1][:7] == ["phmmer", "--noali", "--notextw", "--cpu", "2", "-E", "10"] and "--max" in calls[-1]
    names, aligned = msa_tools.generate_alignment({"1": ["MKV", "MKVAA", "M"]}, ep=0.25, op=2.0)
    assert names == ["1_0", "1_1", "1_2"] and aligned == ["MKV--", "MKVAA", "M----"]
    assert calls[-1][:10] == ["mafft", "--thread", "8", "--maxiterate", "1000", "--globalpair", "--ep", "0.25", "--op", "2.0"]
    assert msa_tools.add_to_msa(["MKVA", "MRVA"], "MKVAA") == ["MKVAA", "MKVA-", "MRVA-"]   # new sequence moved to the top
    assert calls[-1][:2] == ["muscle", "-profile"]
    monkeypatch.setattr(subprocess, "run", lambda argv, **kw: Res("", rc=1, as_bytes=True))
    with pytest.raises(Exception, match="mafft failed"):
        msa_tools.generate_alignment({"1": ["MKV"]})
    monkeypatch.setattr(subprocess, "run", lambda argv, **kw: Res("boom", rc=2))
    with pytest.raises(SystemExit):
        msa_tools.run_phmmer("MKVA", tmp_path / "db.fasta")


class _RecordingSampler:
    """generate_single_batch stand-in: returns the first row of every alignment upper-cased and records the calls."""
    shard_over_ranks = False

    def __init__(self, fail_on_call=None):
        self.calls, self.fail_on_call = [], fail_on_call

    def generate_single_batch(self, msas, **kw):
        self.calls.append((len(msas), kw["target_index"], kw["max_batch"]))
        if self.fail_on_call is not None and len(self.calls) - 1 == self.fail_on_call:
            raise RuntimeError("out of memory (simulated)")
        return [m[kw["target_index"]] for m in msas]


def _run_pgen_msa_revised(tmp_path, monkeypatch, sampler, n_templates=5, seqs_per_template=2, template_batch=4, output=None):
    from _standin import fake_generate_alignment, fake_run_phmmer
    monkeypatch.setattr(pgen_msa_revised, "run_phmmer", fake_run_phmmer)
    monkeypatch.setattr(pgen_msa_revised, "generate_alignment", fake_generate_alignment)
    t, r, o = tmp_path / "t.fasta", tmp_path / "r.fasta", tmp_path / "o.fasta"
    t.write_text("".join(">t%d\nMKV%sLA\n" % (i, "ACDEF"[i % 5]) for i in range(n_templates)))
    r.write_text(">a\nMKVALA\n>b\nMKVCLA\n>c\nMRVDLA\n")
    pgen_msa_revised.pgen_msa(str(t), str(r), output or str(o), seqs_per_template, True, 3, 1, 1, "cuda:0", "esm_msa1", 3, 0.0, 1.53, 1,
                              legacy=True, sampler=sampler, template_batch=template_batch)
    return o


def test_pgen_msa_revised_writes_chunk_by_chunk(tmp_path, monkeypatch):
    """the reference prints every sequence as soon as it exists (pgen_msa_revised.py:107-115, flush=True).  The
    batched pipeline appends and flushes after every chunk of `template_batch` jobs, in the reference's order."""
    s = _RecordingSampler()
    o = _run_pgen_msa_revised(tmp_path, monkeypatch, s)
    assert [c[0] for c in s.calls] == [4, 4, 2] and all(c[1] == -1 and c[2] == 4 for c in s.calls)
    names = [ln[1:] for ln in o.read_text().splitlines() if ln.startswith(">")]
    assert names == ["%d_t%d" % (i, t) for t in range(5) for i in range(2)]


def test_pgen_msa_revised_keeps_finished_chunks_when_a_late_one_fails(tmp_path, monkeypatch):
    s = _RecordingSampler(fail_on_call=2)
    with pytest.raises(RuntimeError):
        _run_pgen_msa_revised(tmp_path, monkeypatch, s)
    names = [ln[1:] for ln in (tmp_path / "o.fasta").read_text().splitlines() if ln.startswith(">")]
    assert names == ["%d_t%d" % (i, t) for t in range(4) for i in range(2)]      # two chunks of four were flushed


def test_pgen_msa_revised_every_rank_writes_unless_one_job_is_sharded(monkeypatch):
    """a process group that merely exists must not silence ranks > 0 (each runs its OWN job by default).  ADVICE r04:
    ...and they must not all truncate and interleave into ONE file when they were handed the same path: ranks > 0 write
    `<path>.rank<r>`.  A job sharded over the ranks is written by rank 0 alone."""
    import torch.distributed as dist
    from protein_gibbs_sampler_amd import sharding

    class Ctx:
        rank, world = 1, 2
    monkeypatch.setattr(sharding, "dist_context", lambda: Ctx())
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda *a: 2)
    monkeypatch.setattr(dist, "get_rank", lambda *a: Ctx.rank)
    s = _RecordingSampler()
    with pytest.warns(UserWarning, match="rank 1"):
        assert pgen_msa_revised._output_path_for_this_process(s, "o.fasta") == "o.fasta.rank1"     # rank 1, sharding off: its own file
    s.shard_over_ranks = True
    assert pgen_msa_revised._output_path_for_this_process(s, "o.fasta") is None                   # rank 1 of a sharded job: rank 0 writes
    Ctx.rank = 0
    assert pgen_msa_revised._output_path_for_this_process(s, "o.fasta") == "o.fasta"
    s.shard_over_ranks = False
    assert pgen_msa_revised._output_path_for_this_process(s, "o.fasta") == "o.fasta"


def test_pgen_msa_revised_bad_output_path_leaves_no_scratch_fasta(tmp_path, monkeypatch):
    """the output file is opened inside the try block -- a path that cannot be opened still removes the scratch
    reference FASTA in the finally."""
    import glob
    import tempfile
    monkeypatch.setattr(tempfile, "tempdir", str(tmp_path))
    before = set(glob.glob(str(tmp_path / "tmp*")))
    with pytest.raises(OSError):
        _run_pgen_msa_revised(tmp_path, monkeypatch, _RecordingSampler(), output=str(tmp_path / "no_such_dir" / "o.fasta"))
    assert set(glob.glob(str(tmp_path / "tmp*"))) == before
