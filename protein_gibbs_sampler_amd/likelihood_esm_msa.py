#!/usr/bin/env python3
"""Average log-likelihood of query sequences under ESM-MSA-1b given a reference alignment, with the surface of the
reference's `likelihood_esm_msa.py` (/root/reference/src/pgen/likelihood_esm_msa.py:19-146 `main`, :150-226 flags).

Per query the context MSA is: the caller's own MSA (`in_msas`), or the query on top of a subset of the reference
alignment (`random` / `in_order`, optionally redrawn per query, optionally aligned in with muscle), or a fresh mafft
alignment of the query's best phmmer hits (`top_hits`).  Columns that are gaps in the query are dropped before scoring;
scores are `log_likelihood_batch(..., count_gaps=False)` of row 0."""
import argparse
import os
import sys
import tempfile
import textwrap
import warnings

from . import models
from ._cli import RawAndDefaultsFormatter, add_engine_args
from .esm_msa_sampler import ESM_MSA_sampler
from .fasta_io import SequenceSubsetter, parse_fasta, write_sequential_fasta
from .msa_tools import add_to_msa, delete_msa_cols, generate_alignment, run_phmmer  # noqa: F401

POSITIONAL_SCORE_SEP = ";"


def main(input_h, output_h, masking_off, sampler, reference_msa_handle=None, in_msas=None, delete_insertions=False, batch_size=1,
         subset_strategy="random", alignment_size=sys.maxsize, subset_random_seed=None, redraw=False, unaligned_queries=False,
         mask_distance=float("inf"), csv=False, positionwise=None, keep_identical=False):
    clean_flag = "delete" if delete_insertions else "upper"
    sep = "," if csv else "\t"
    print(f"id{sep}esm-msa", file=output_h)
    positionwise_h = None
    if positionwise is not None:
        positionwise_h = open(positionwise, "w")
        print(f"id{sep}esm-msa", file=positionwise_h)

    state = {"seed": subset_random_seed, "fixed": None}
    reference_db_path, renamed = None, {}
    if not in_msas:
        reference_msa = parse_fasta(reference_msa_handle, clean=clean_flag)
        if subset_strategy == "top_hits":              # a fresh search + alignment for every query
            with tempfile.NamedTemporaryFile(delete=False, mode="w") as tmp:
                write_sequential_fasta(tmp, reference_msa)
                reference_db_path = tmp.name
            renamed = dict(zip(*parse_fasta(reference_db_path, return_names=True)))
        else:
            state["fixed"] = SequenceSubsetter.subset(reference_msa, alignment_size, strategy=subset_strategy, random_seed=state["seed"])

    in_seqs = dict(zip(*parse_fasta(input_h, return_names=True, clean=clean_flag)))

    def context_msa(name):
        seq = in_seqs[name]
        if in_msas:
            return in_msas[name]
        if subset_strategy == "top_hits":
            rows = [seq]
            for hit in run_phmmer(seq, reference_db_path):
                if len(rows) == alignment_size:
                    break
                if renamed[hit] != seq or keep_identical:
                    rows.append(renamed[hit])
            if len(rows) < alignment_size:
                warnings.warn(f"Warning: fewer than {alignment_size - 1} hits found for template seq {name}")
            return generate_alignment({"1": rows})[1]
        rows = state["fixed"]
        if redraw:
            rows = SequenceSubsetter.subset(reference_msa, alignment_size, strategy=subset_strategy, random_seed=state["seed"])
            if state["seed"] is not None:
                state["seed"] += 1000000
        return add_to_msa(rows, seq) if unaligned_queries else [seq] + list(rows)

    try:
        names = list(in_seqs.keys())
        for start in range(0, len(names), batch_size):
            chunk = names[start:start + batch_size]
            msas = []
            for name in chunk:
                msa = context_msa(name)
                msas.append(delete_msa_cols(msa, [i for i, c in enumerate(msa[0]) if c == "-"]))
            scores = sampler.log_likelihood_batch(msas, with_masking=not masking_off, count_gaps=False, mask_distance=mask_distance,
                                                  batch_size=batch_size)
            for name, (score, positional) in zip(chunk, scores):
                print(f"{name}{sep}{score}", file=output_h)
                if positionwise_h is not None:
                    print(f"{name}{sep}{POSITIONAL_SCORE_SEP.join(str(round(x, 3)) for x in positional)}", file=positionwise_h)
            output_h.flush()
            if positionwise_h is not None:
                positionwise_h.flush()
    finally:
        if positionwise_h is not None:
            positionwise_h.close()
        if reference_db_path is not None:
            os.unlink(reference_db_path)


def build_parser():
    parser = argparse.ArgumentParser(description=textwrap.dedent("""Calculates average log likelihood of a fasta from the ESM-MSA model.

    writes a tab separated output file with columns:
    sequence name, score
    """), formatter_class=RawAndDefaultsFormatter)
    parser.add_argument("-o", type=str, default=None, help="output table (default: stdout)")
    parser.add_argument("-i", default=None, help="A fasta file with sequences to calculate log likelihood for")
    parser.add_argument("--reference_msa", default=None, required=True,
                        help="fasta with the reference msa (for subset_strategy top_hits: unaligned reference sequences).")
    parser.add_argument("--device", type=str, default="gpu", help="gpu (cuda:0) or cuda:[int]")
    parser.add_argument("--masking_off", action="store_true", default=False, help="If set, no masking is done.")
    parser.add_argument("--delete_insertions", action="store_true", default=False,
                        help="remove all lowercase and '.' characters from input sequences. Default: lower -> upper and '.' -> '-'.")
    parser.add_argument("--alignment_size", type=int, default=sys.maxsize,
                        help="sample this many sequences from the reference alignment (recommended 31-255). Default: all of it.")
    parser.add_argument("--keep_identical", action="store_true", default=False,
                        help="top_hits: keep hits identical to the query (thrown out by default).")
    parser.add_argument("--batch_size", type=int, default=1, help="msa instances per forward.")
    parser.add_argument("--subset_strategy", default="random", choices=["in_order", "random", "top_hits"],
                        help="random: draw randomly, in_order: first sequences of the reference alignment, top_hits: phmmer per query "
                             "against the reference sequences and a MAFFT MSA of the top hits.")
    parser.add_argument("--subset_random_seed", default=None, type=int, help="seed of the random subsetter (+1000000 after each draw).")
    parser.add_argument("--redraw", action="store_true", default=False, help="random: a new draw of reference sequences for each query.")
    parser.add_argument("--unaligned_queries", action="store_true", default=False,
                        help="queries are unaligned / from another alignment: add each to the reference alignment with muscle -profile.")
    parser.add_argument("--mask_distance", type=int, default=None,
                        help="mask several positions per copy, (mask_distance - 1) unmasked positions apart. Default: one at a time.")
    parser.add_argument("--csv", action="store_true", default=False, help="If set, then outputs will be csv files.")
    parser.add_argument("--positionwise", type=str, default=None, help="also write per-position log likelihoods (';' separated) to this file.")
    add_engine_args(parser)
    return parser


def cli(argv=None):
    args = build_parser().parse_args(argv)
    if args.redraw and args.subset_strategy == "in_order":
        raise ValueError("redraw is set, but subset_strategy is 'in_order', so all the draws will be the same. "
                         "That's probably not what you're trying to do.")
    mask_distance = float("inf") if args.mask_distance is None else args.mask_distance
    if mask_distance < 1:
        raise ValueError("mask distance must be an integer >= 1.")
    sampler = ESM_MSA_sampler(models.ESM_MSA1(checkpoint=args.checkpoint, precision=args.precision, synthetic=args.synthetic_weights), device=args.device)
    input_handle = open(args.i) if args.i is not None else sys.stdin
    output_handle = open(args.o, "w") if args.o is not None else sys.stdout
    try:
        with open(args.reference_msa) as reference_msa_handle:
            main(input_h=input_handle, output_h=output_handle, masking_off=args.masking_off, sampler=sampler,
                 reference_msa_handle=reference_msa_handle, delete_insertions=args.delete_insertions, batch_size=args.batch_size,
                 subset_strategy=args.subset_strategy, alignment_size=args.alignment_size, subset_random_seed=args.subset_random_seed,
                 redraw=args.redraw, unaligned_queries=args.unaligned_queries, mask_distance=mask_distance, csv=args.csv,
                 positionwise=args.positionwise, keep_identical=args.keep_identical)
    finally:
        if args.i is not None:
            input_handle.close()
        if args.o is not None:
            output_handle.close()


if __name__ == "__main__":
    cli()
