#!/usr/bin/env python3
"""Command-line front end with the surface of the reference's `pgen_esm.py` (/root/reference/src/pgen/pgen_esm.py):
TSV lines `name <TAB> dict-of-sampler-arguments` -> `<out>/<name>.fasta` (+ `<out>/specification.tsv` echo)."""
import argparse
import sys
import textwrap
from pathlib import Path

from . import models
from ._cli import RawAndDefaultsFormatter, add_engine_args, parse_line_args, seed_everything
from .esm_sampler import ESM_sampler
from .fasta_io import write_sequential_fasta

model_map = {"esm1b": models.ESM1b, "esm1v": models.ESM1v, "esm6": models.ESM6, "esm12": models.ESM12, "esm34": models.ESM34}


def main(input_h, output_p, args):
    sampler = ESM_sampler(model_map[args.model](checkpoint=args.checkpoint, precision=args.precision, synthetic=args.synthetic_weights), device=args.device)
    with open(output_p / "specification.tsv", "w") as output_h:
        for line in input_h:
            line = line.strip()
            if not line:
                continue
            fields = line.split("\t")
            if len(fields) != 2:
                print(f"Expected 2 values in specification file (name, line_args), got {len(fields)}")
                print("\t".join(fields))
                continue
            print("\t".join(fields))
            print("\t".join(fields), file=output_h)
            name, line_args = fields[0], parse_line_args(fields[1])
            sequences = sampler.generate(args.num_output_sequences, batch_size=args.batch_size, **line_args)
            write_sequential_fasta(output_p / (name + ".fasta"), sequences)


def build_parser():
    parser = argparse.ArgumentParser(
        description=textwrap.dedent("""Samples from an ESM BERT model to generate new protein sequences (MI355X engine).

            Input should be a tab separated file where columns are:
            sample name, dict of sampler arguments
            """),
        epilog=textwrap.dedent("""
            Available sampler arguments: seed_seq, in_order, max_len, leader_length, leader_length_percent, top_k, temperature,
            num_iters, burnin, mask, num_positions, num_positions_percent, indexes, rollover_from_start
            (see ESM_sampler.generate)."""),
        formatter_class=RawAndDefaultsFormatter)
    parser.add_argument("-o", default=".", help="a directory to save the outputs to.")
    parser.add_argument("-i", default=None, help="tab separated file: [sample name] \\t [dict of arguments for the sampler].")
    parser.add_argument("--batch_size", type=int, default=1, help="batch size for sampling (sequences per iteration).")
    parser.add_argument("--num_output_sequences", type=int, default=1, help="total number of sequences to generate.")
    parser.add_argument("--device", type=str, default="gpu", help="gpu (cuda:0) or cuda:[int]; cpu is accepted by the grammar "
                        "but sampling needs an MI355X")
    parser.add_argument("--model", type=str, default="esm1b", choices=sorted(model_map), help="which model to use")
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
