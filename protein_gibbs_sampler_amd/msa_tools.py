"""Host-side alignment helpers either side of the MSA Gibbs path: gap bookkeeping, column filtering and the wrappers
around the external aligner / homology-search programs.

Mirrors (behaviour, not text) /root/reference/src/pgen/utils.py:42-85 (`unalign`, `add_gaps_back`), :171-208
(`add_to_msa`, muscle), :231-263 (`write_partitioned_fasta`, `generate_alignment`, mafft), :265-314 (`run_phmmer`)
and /root/reference/src/pgen/pgen_msa_revised.py:16-47 (`delete_msa_cols`, `count_gaps_per_column`,
`apply_gap_threshold`).  mafft / muscle / phmmer stay subprocesses exactly as in the reference; the reference reads
phmmer's text report through Biopython's SearchIO, here the hit table is parsed directly (`parse_phmmer_hits`) so the
module has no dependency beyond the standard library.
"""
import os
import string
import subprocess
import sys
import tempfile

from .fasta_io import parse_fasta_string, write_sequential_fasta


def unalign(sequence):
    """(letters of `sequence` upper-cased, gap mask): the mask has None where a letter was kept and the original
    character ('.', '*', '-', digits ...) everywhere else, so `add_gaps_back` can restore the layout."""
    kept, gap_mask = [], []
    for c in sequence.upper():
        if c in string.ascii_uppercase:
            kept.append(c)
            gap_mask.append(None)
        else:
            gap_mask.append(c)
    return "".join(kept), gap_mask


def add_gaps_back(sequence, gap_mask):
    """Inverse of `unalign`: None slots of the mask are filled, in order, from `sequence`."""
    it = iter(sequence)
    return "".join(next(it) if c is None else c for c in gap_mask)


def write_partitioned_fasta(path, sequences):
    """`sequences`: {category: [seq, ...]} -> records named `<category>_<i>`."""
    with open(path, "w") as out:
        for category, seqs in sequences.items():
            for i, seq in enumerate(seqs):
                print(f">{category}_{i}\n{seq}", file=out)


def delete_msa_cols(msa, cols):
    """The alignment without the columns whose 0-based indices are in `cols`."""
    drop = set(cols)
    return ["".join(c for i, c in enumerate(seq) if i not in drop) for seq in msa]


def count_gaps_per_column(msa):
    """Number of '-' per column (width = the first row's)."""
    return [sum(seq[i] == "-" for seq in msa) for i in range(len(msa[0]))]


def apply_gap_threshold(msa, gap_threshold):
    """Indices of the columns in which MORE than `gap_threshold` percent of the rows are gaps."""
    bound = len(msa) * (gap_threshold / 100)
    return [i for i, n in enumerate(count_gaps_per_column(msa)) if n > bound]


def _run_tool(argv, **kw):
    try:
        return subprocess.run(argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)
    except FileNotFoundError:
        raise Exception(f"external program '{argv[0]}' not found on PATH (it is run as a subprocess, as in the reference)") from None


def generate_alignment(sequences, ep=0.0, op=1.53):
    """mafft G-INS-i alignment of {category: [seq, ...]}; returns (names, aligned sequences) in input order."""
    with tempfile.TemporaryDirectory() as tmp:
        fasta = os.path.join(tmp, "tmp.fasta")
        write_partitioned_fasta(fasta, sequences)
        res = _run_tool(["mafft", "--thread", "8", "--maxiterate", "1000", "--globalpair", "--ep", str(ep), "--op", str(op), fasta])
    if res.returncode != 0:
        print(res.stderr, file=sys.stderr)
        raise Exception("mafft failed")
    return parse_fasta_string(res.stdout.decode("utf-8"), True)


def add_to_msa(msa, new_seq):
    """muscle -profile: `new_seq` aligned against `msa`; the new sequence comes first in the returned list."""
    tag = "new_seq"
    with tempfile.TemporaryDirectory() as tmp:
        p1, p2 = os.path.join(tmp, "out1.fasta"), os.path.join(tmp, "out2.fasta")
        write_sequential_fasta(p1, msa)
        with open(p2, "w") as f:
            print(f">{tag}\n{new_seq}", file=f)
        res = _run_tool(["muscle", "-profile", "-in1", p1, "-in2", p2], encoding="utf-8")
    names, seqs = parse_fasta_string(res.stdout, True)
    if tag not in names:
        print(names, res.stdout, res.stderr, sep="\n", file=sys.stderr)
        raise ValueError(f"'{tag}' is not in list")
    seq = seqs.pop(names.index(tag))
    return [seq] + seqs


def parse_phmmer_hits(report):
    """Target names of the per-sequence hit table of a HMMER3 text report, best hit first (rows below the inclusion
    threshold included, as SearchIO's hit list has them)."""
    hits, in_table = [], False
    for line in report.splitlines():
        s = line.strip()
        if s.startswith("Scores for complete sequence"):
            in_table = True
            continue
        if not in_table:
            continue
        if s.startswith("Domain annotation") or s.startswith("Internal pipeline") or s.startswith("//"):
            break
        if not s or s.startswith("---") or s.startswith("E-value") or s.startswith("------ inclusion") or s.startswith("[No hits"):
            continue
        fields = s.split()
        if len(fields) >= 9:
            try:
                float(fields[0]), float(fields[1])
            except ValueError:
                continue
            hits.append(fields[8])
    return hits


def run_phmmer(query, database, evalue=10, cpu=2, max_mode=False):
    """phmmer of one protein sequence against a FASTA database: hit names ranked best first."""
    with tempfile.TemporaryDirectory() as tmp:
        qpath = os.path.join(tmp, "query.fa")
        with open(qpath, "w") as f:
            print(f">QUERY\n{query}", file=f)
        argv = ["phmmer", "--noali", "--notextw", "--cpu", str(cpu), "-E", str(evalue)] + (["--max"] if max_mode else []) + [qpath, str(database)]
        res = _run_tool(argv, encoding="utf-8")
    if res.returncode != 0:
        print(f"Error in hmmer execution: \n{res.stdout}\n{res.stderr}", file=sys.stderr)
        sys.exit(1)
    return parse_phmmer_hits(res.stdout)
