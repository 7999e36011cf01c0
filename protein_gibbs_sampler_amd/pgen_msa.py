#!/usr/bin/env python3
"""Command-line front end with the surface of the reference's `pgen_msa.py` (/root/reference/src/pgen/pgen_msa.py):
TSV lines `name <TAB> dict-of-sampler-arguments <TAB> seed-msa.fasta` -> `<out>/<name>.fasta` (+ specification.tsv)."""
import argparse
import math
import sys
import textwrap
from pathlib import Path

from . import models
from ._cli import RawAndDefaultsFormatter, add_engine_args, parse_line_args, seed_everything
from .esm_msa_sampler import ESM_MSA_sampler
from .fasta_io import SequenceSubsetter, parse_fasta, write_sequential_fasta

model_map = {"esm_msa1": models.ESM_MSA1}


def main(input_h, output_p, args):
    clean_flag = "delete" if args.delete_insertions else "upper"
    sampler = ESM_MSA_sampler(model_map[args.model](checkpoint=args.checkpoint, precision=args.precision, synthetic=args.synthetic_weights), device=args.device)
    with open(output_p / "specification.tsv", "w") as output_h:
        for line in input_h:
            line = line.strip()
            if not line:
                continue
            fields = line.split("\t")
            if len(fields) != 3:
                print(f"Expected 3 values in specification file (name, line_args, input_msa), got {len(fields)}")
                print("\t".join(fields))
                continue
            print("\t".join(fields))
            print("\t".join(fields), file=output_h)
            name, line_args = fields[0], parse_line_args(fields[1])
            input_msa = parse_fasta(fields[2], clean=clean_flag)
            alignment_size = len(input_msa) if args.alignment_size == sys.maxsize else args.alignment_size
            sequences = []
            for _ in range(math.ceil(args.num_output_sequences / alignment_size)):
                batch_msa = SequenceSubsetter.subset(input_msa, alignment_size, args.keep_first_sequence, args.subset_strategy)
                sequences += sampler.generate(n_samples=len(batch_msa), seed_msa=batch_msa, batch_size=args.batch_size,
                                              show_progress_bar=False, **line_args)
            write_sequential_fasta(output_p / (name + ".fasta"), sequences[0:args.num_output_sequences])


def build_parser():
    parser = argparse.ArgumentParser(
        description=textwrap.dedent("""Samples from the ESM-MSA model to generate new protein sequences (MI355X engine).

            Input should be a tab separated file where columns are:
            sample name, dict of sampler arguments, fasta of seed sequences
            """),
        epilog="Available sampler arguments: see ESM_MSA_sampler.generate.", formatter_class=RawAndDefaultsFormatter)
    parser.add_argument("-o", default=".", help="a directory to save the outputs in")
    parser.add_argument("-i", default=None, help="tab separated file: [sample name] \\t [dict of arguments] \\t [seed msa, fasta or a2m].")
    parser.add_argument("--batch_size", type=int, default=1, help="MSA instances per iteration (the reference CLI only allows 1; "
                        "the engine batches any number)")
    parser.add_argument("--num_output_sequences", type=int, default=1, help="total number of sequences to generate.")
    parser.add_argument("--device", type=str, default="gpu", help="gpu (cuda:0) or cuda:[int]")
    parser.add_argument("--model", type=str, default="esm_msa1", choices=sorted(model_map), help="which model to use")
    parser.add_argument("--delete_insertions", action="store_true", default=False,
                        help="remove all lowercase and '.' characters from input sequences. Default: lower -> upper and '.' -> '-'.")
    parser.add_argument("--alignment_size", type=int, default=sys.maxsize,
                        help="sample this many sequences from the input alignment before sampling (recommended 32-256). "
                             "Default: the entire input alignment.")
    parser.add_argument("--keep_first_sequence", action="store_true", default=False,
                        help="keep the first sequence and sample the rest according to subset_strategy.")
    parser.add_argument("--subset_strategy", default="random", choices=sorted(SequenceSubsetter.subset_strategies),
                        help="how to subset the input alignment to get it to the desired size.")
    add_engine_args(parser)
    return parser


def cli(argv=None):
    args = build_parser().parse_args(argv)
    seed_everything(args.seed)
    output_path = Path(args.o)
    output_path.mkdir(exist_ok=True)
    if args.i is not None:
        with open(args.i) as handle:
            main(handle, output_path, args)
    else:
        main(sys.stdin, output_path, args)


if __name__ == "__main__":
    cli()
