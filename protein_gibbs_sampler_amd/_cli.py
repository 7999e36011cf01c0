"""Shared pieces of the command-line front ends (pgen_esm.py / pgen_msa.py)."""
import argparse
import ast


class RawAndDefaultsFormatter(argparse.ArgumentDefaultsHelpFormatter, argparse.RawTextHelpFormatter):
    pass


def parse_line_args(text):
    """The second TSV column: a Python dict literal of sampler keyword arguments.

    The reference passes this column to eval() (src/pgen/pgen_esm.py:25).  Here it is parsed as a literal
    (ast.literal_eval) with `inf` / `float('inf')` accepted, which covers every form in the reference's README and
    examples without executing arbitrary code -- a deliberate, documented deviation."""
    src = text.replace("float('inf')", "1e999").replace('float("inf")', "1e999")
    tree = ast.parse(src.strip(), mode="eval")
    for node in ast.walk(tree):
        if isinstance(node, ast.Name) and node.id in ("inf", "Infinity"):
            node.__class__ = ast.Constant
            node.value = float("inf")
            node.kind = None
    out = ast.literal_eval(tree)
    if not isinstance(out, dict):
        raise ValueError("sampler arguments must be a dict literal, got: " + text)
    return out


def add_engine_args(parser):
    parser.add_argument("--checkpoint", default=None, help="fair-esm .pt checkpoint to load (default: torch hub cache; it is an "
                        "error if none is found)")
    parser.add_argument("--synthetic-weights", dest="synthetic_weights", action="store_true",
                        help="opt in to seeded random weights of the model's architecture when no checkpoint is available "
                             "(benchmarks / plumbing tests: the output is not biologically meaningful)")
    parser.add_argument("--precision", default="auto", choices=["auto", "bf16", "fp16", "fp32"],
                        help="auto = fp16 operands with a range guard (weights scanned, one probe forward, non-finite logits "
                             "detected per call: falls back to bf16 with one warning); bf16 = the benchmarked throughput mode; "
                             "fp16 = the same kernels with fp16 operands, no fallback (8x smaller logit error than bf16, ~3 %% "
                             "slower); fp32 = parity mode (split-bf16 GEMMs and attention, logits within 1e-3 of fp32)")
    parser.add_argument("--seed", type=int, default=None, help="seed random and torch (positions and token draws) for reproducible output")


def seed_everything(seed):
    if seed is not None:
        import random
        import torch
        random.seed(seed)
        torch.manual_seed(seed)
