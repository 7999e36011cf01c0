#!/usr/bin/env python3
"""FASTA cleaner with the surface of the reference's `clean_fasta.py` (/root/reference/src/pgen/clean_fasta.py):
re-writes a FASTA/a2m through one of parse_fasta's cleaning modes."""
import argparse
import sys

from .fasta_io import parse_fasta


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("-i", default=None)
    parser.add_argument("-o", default=None)
    parser.add_argument("--clean_strategy", type=str, default=None, choices=["delete", "upper", "unalign"], required=True, help="")
    parser.add_argument("--full_name", action="store_true", default=False, help="keep the whole header line, including the description")
    args = parser.parse_args(argv)
    src = open(args.i) if args.i is not None else sys.stdin
    dst = open(args.o, "w") if args.o is not None else sys.stdout
    try:
        names, seqs = parse_fasta(src, return_names=True, clean=args.clean_strategy, full_name=args.full_name)
        for name, seq in zip(names, seqs):
            print(f">{name}\n{seq}", file=dst)
    finally:
        if args.i is not None:
            src.close()
        if args.o is not None:
            dst.close()


if __name__ == "__main__":
    main()
