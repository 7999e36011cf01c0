#!/usr/bin/env python3
"""Front end with the surface of the reference's `pgen_esm_from_fasta.py` (/root/reference/src/pgen/pgen_esm_from_fasta.py):
TSV lines `name <TAB> dict-of-sampler-arguments <TAB> seeds.fasta`; every output sequence starts from a seed drawn with
`random.choice` from the FASTA, un-aligned first (gaps optionally re-inserted afterwards) -> `<out>/<name>.fasta`."""
import argparse
import random
import sys
import textwrap
from pathlib import Path

from . import models
from ._cli import RawAndDefaultsFormatter, add_engine_args, parse_line_args, seed_everything
from .esm_sampler import ESM_sampler
from .fasta_io import parse_fasta, write_sequential_fasta
from .msa_tools import add_gaps_back, unalign

model_map = {"esm1b": models.ESM1b, "esm6": models.ESM6, "esm12": models.ESM12, "esm34": models.ESM34}


def main(input_h, output_p, args, sampler=None):
    if sampler is None:
        sampler = ESM_sampler(model_map[args.model](checkpoint=getattr(args, "checkpoint", None),
                                                    precision=getattr(args, "precision", "auto"),
                                                    synthetic=getattr(args, "synthetic_weights", False)), device=args.device)
    with open(output_p / "specification.tsv", "w") as output_h:
        for line in input_h:
            fields = line.strip().split("\t")
            if len(fields) != 3:                       # blank and malformed lines are skipped silently, as in the reference
                continue
            print("\t".join(fields))
            print("\t".join(fields), file=output_h)
            name, line_args = fields[0], parse_line_args(fields[1])
            seeds = parse_fasta(fields[2], clean=None)
            sequences = []
            for _ in range(args.num_output_sequences):
                seed, gap_mask = unalign(random.choice(seeds))
                generated = sampler.generate(n_samples=1, seed_seq=seed, batch_size=args.batch_size, show_progress_bar=False,
                                             **line_args)[0]
                sequences.append(add_gaps_back(generated, gap_mask) if args.keep_gap_positions else generated)
            write_sequential_fasta(output_p / (name + ".fasta"), sequences)


def build_parser():
    parser = argparse.ArgumentParser(
        description=textwrap.dedent("""Samples from an ESM BERT model to generate new protein sequences.

            Input should be a tab separated file where columns are:
            sample name, dict of sampler arguments, fasta of seed sequences
            """),
        epilog="Available sampler arguments: see ESM_sampler.generate.", formatter_class=RawAndDefaultsFormatter)
    parser.add_argument("-o", default=".", help="a directory to save the outputs to.")
    parser.add_argument("-i", default=None, help="tab separated file: [sample name] \\t [dict of arguments for the sampler] \\t [path to fasta file].")
    parser.add_argument("--batch_size", type=int, default=1, choices={1}, help="batch size for sampling (sequences per iteration). Must be 1.")
    parser.add_argument("--num_output_sequences", type=int, default=1, help="total number of sequences to generate.")
    parser.add_argument("--device", type=str, default="gpu", help="gpu (cuda:0) or cuda:[int]")
    parser.add_argument("--model", type=str, default="esm1b", choices=sorted(model_map), help="which model to use")
    parser.add_argument("--keep_gap_positions", action="store_true", default=False,
                        help="remember where the gaps are in the seed and put them back into the generated sequence.")
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
