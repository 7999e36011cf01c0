#!/usr/bin/env python3
"""Average log-likelihood of FASTA sequences under an ESM masked LM, with the surface of the reference's
`likelihood_esm.py` (/root/reference/src/pgen/likelihood_esm.py:15-57 loop and table format, :60-104 flags):
output is `id<sep>score` rows (tab or comma), optionally a second table with the ';'-joined per-position
log-likelihoods rounded to 3 decimals."""
import argparse
import sys
import textwrap

from . import models
from ._cli import RawAndDefaultsFormatter, add_engine_args
from .esm_sampler import ESM_sampler
from .fasta_io import parse_fasta

POSITIONAL_SCORE_SEP = ";"
model_map = {"esm1b": models.ESM1b, "esm6": models.ESM6, "esm12": models.ESM12, "esm34": models.ESM34, "esm1v": models.ESM1v}


def main(input_h, output_h, masking_off, device, model, batch_size, mask_distance, csv, score_name, positionwise=None, sampler=None):
    if sampler is None:
        sampler = ESM_sampler(model_map[model](), device=device)
    records = list(zip(*parse_fasta(input_h, return_names=True, clean="unalign")))
    sep = "," if csv else "\t"
    if score_name is None:
        score_name = model
    positionwise_h = open(positionwise, "w") if positionwise is not None else None
    try:
        print(f"id{sep}{score_name}", file=output_h)
        if positionwise_h is not None:
            print(f"id{sep}{score_name}", file=positionwise_h)
        for start in range(0, len(records), batch_size):
            chunk = records[start:start + batch_size]
            scores = sampler.log_likelihood_batch([seq for _, seq in chunk], with_masking=not masking_off,
                                                  mask_distance=mask_distance, batch_size=batch_size)
            for (name, _), (score, positional) in zip(chunk, scores):
                print(f"{name}{sep}{score}", file=output_h)
                if positionwise_h is not None:
                    print(f"{name}{sep}{POSITIONAL_SCORE_SEP.join(str(round(x, 3)) for x in positional)}", file=positionwise_h)
            output_h.flush()
            if positionwise_h is not None:
                positionwise_h.flush()
    finally:
        if positionwise_h is not None:
            positionwise_h.close()


def build_parser():
    parser = argparse.ArgumentParser(description=textwrap.dedent("""Calculates average log likelihood of a fasta ESM BERT model.

    writes a tab separated output file with columns:
    sequence name, score
    """), formatter_class=RawAndDefaultsFormatter)
    parser.add_argument("-o", type=str, default=None, help="output table (default: stdout)")
    parser.add_argument("-i", default=None, help="A fasta file with sequences to score. Gaps and stop codons are removed first.")
    parser.add_argument("--batch_size", type=int, default=1, help="How many sequences to batch together.")
    parser.add_argument("--device", type=str, default="gpu", help="gpu (cuda:0) or cuda:[int]")
    parser.add_argument("--masking_off", action="store_true", default=False, help="If set, no masking is done.")
    parser.add_argument("--mask_distance", type=int, default=None,
                        help="mask several positions per copy, (mask_distance - 1) unmasked positions apart. Default: one position at a time.")
    parser.add_argument("--model", type=str, default="esm1v", choices=sorted(model_map), help="Which model to use.")
    parser.add_argument("--csv", action="store_true", default=False, help="If set, then output will be a csv file.")
    parser.add_argument("--score_name", type=str, default=None, help="name of the second column (default: the model name).")
    parser.add_argument("--positionwise", type=str, default=None, help="also write per-position log likelihoods (';' separated) to this file.")
    add_engine_args(parser)
    return parser


def cli(argv=None):
    args = build_parser().parse_args(argv)
    mask_distance = float("inf") if args.mask_distance is None else args.mask_distance
    if mask_distance < 1:
        raise ValueError("mask distance must be an integer >= 1.")
    if args.masking_off and args.mask_distance is not None:
        raise ValueError("--masking_off and --mask_distance are both set, that doesn't make sense.")
    sampler = ESM_sampler(model_map[args.model](checkpoint=args.checkpoint, precision=args.precision, synthetic=args.synthetic_weights), device=args.device)
    input_handle = open(args.i) if args.i is not None else sys.stdin
    output_handle = open(args.o, "w") if args.o is not None else sys.stdout
    try:
        main(input_handle, output_handle, args.masking_off, args.device, args.model, args.batch_size, mask_distance, args.csv,
             args.score_name, args.positionwise, sampler=sampler)
    finally:
        if args.i is not None:
            input_handle.close()
        if args.o is not None:
            output_handle.close()


if __name__ == "__main__":
    cli()
