#!/usr/bin/env python3
"""Template-driven sequence generation around `ESM_MSA_sampler.generate_single`, with the surface of the reference's
`pgen_msa_revised.py` (/root/reference/src/pgen/pgen_msa_revised.py:49-117 pipeline, :119-153 flags):

  for every template: phmmer the template against the reference set -> take the best `alignment_size - 1` hits ->
  mafft them with the template on top -> (default) drop the template's gap columns and exclude columns that are gaps in
  more than `gap_percent_threshold` % of the rows / (--legacy) swap the first and last rows and resample row -1 ->
  `seqs_per_template` calls of generate_single -> `>i_templatename` records, gaps stripped.

phmmer and mafft are subprocesses (msa_tools); the Gibbs passes run on the MI355X engine.
"""
import argparse
import os
import sys
import tempfile
import textwrap
import warnings

from . import models, sharding
from ._cli import RawAndDefaultsFormatter, add_engine_args, seed_everything
from .esm_msa_sampler import ESM_MSA_sampler
from .fasta_io import parse_fasta, write_sequential_fasta
from .msa_tools import apply_gap_threshold, count_gaps_per_column, delete_msa_cols, generate_alignment, run_phmmer  # noqa: F401

model_map = {"esm_msa1": models.ESM_MSA1}


def build_template_alignment(template_name, template_seq, reference_seqs, reference_db_path, alignment_size, keep_identical,
                             ep, op, legacy, gap_percent_threshold, debug):
    """(alignment rows, excluded column indices, row to resample) for one template."""
    rows = [template_seq]
    for hit in run_phmmer(template_seq, reference_db_path, max_mode=debug):
        if len(rows) == alignment_size:
            break
        if reference_seqs[hit] != template_seq or keep_identical:
            rows.append(reference_seqs[hit])
    if len(rows) < alignment_size:
        warnings.warn(f"Warning: fewer than {alignment_size - 1} hits found for template seq {template_name}")
    _, alignment = generate_alignment({"1": rows}, ep=ep, op=op)          # mafft keeps the input order
    if legacy:                                                            # original behaviour: template goes last, row -1 is resampled
        alignment[0], alignment[-1] = alignment[-1], alignment[0]
        return alignment, [], -1
    template_gaps = [i for i, c in enumerate(alignment[0]) if c == "-"]
    alignment = delete_msa_cols(alignment, template_gaps)
    return alignment, apply_gap_threshold(alignment, gap_percent_threshold), 0


def _output_path_for_this_process(sampler, output_path):
    """Where this process writes, or None.  ONE job split over the torch.distributed ranks (sampler.shard_over_ranks /
    PGIBBS_SHARD_OVER_RANKS with world size > 1): every rank holds the full result and rank 0 alone writes `output_path`.  A
    process group that merely exists (each rank sampling its OWN templates) silences nobody -- but ranks > 0 write
    `<output_path>.rank<r>`: handed the same path (the usual torchrun script), all ranks would otherwise truncate and
    interleave into one file, and asking the other ranks what path they were given would need a collective that ranks running
    independent jobs cannot be assumed to join."""
    if sharding.sharding_requested(getattr(sampler, "shard_over_ranks", False)):
        ctx = sharding.dist_context()
        return output_path if ctx is None or ctx.rank == 0 else None
    try:
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 else 0
    except ImportError:
        rank = 0
    if rank == 0:
        return output_path
    warnings.warn(f"rank {rank} of a torch.distributed group without shard_over_ranks: writing {output_path}.rank{rank}")
    return f"{output_path}.rank{rank}"


def pgen_msa(templates_path, references_path, output_path, seqs_per_template, keep_identical, steps, passes, burn_in, device,
             model, alignment_size, ep, op, top_k, legacy=False, gap_percent_threshold=80, debug=False, sampler=None,
             template_batch=4):
    templates = list(zip(*parse_fasta(templates_path, clean="unalign", return_names=True)))
    references = parse_fasta(references_path, clean="unalign")
    if sampler is None:
        sampler = ESM_MSA_sampler(model_map[model](), device=device)
    # the references get sequential names (0..n-1) in a scratch FASTA: that file is phmmer's database and its names
    # are what the hits come back as
    with tempfile.NamedTemporaryFile(delete=False, mode="w") as tmp:
        write_sequential_fasta(tmp, references)
        reference_db_path = tmp.name
    my_output = _output_path_for_this_process(sampler, output_path)
    ctx = sharding.dist_context() if sharding.sharding_requested(getattr(sampler, "shard_over_ranks", False)) else None
    # jobs per generate_single_batch call: `template_batch` alignments share a forward; a sharded run hands every rank that many
    chunk_jobs = max(1, template_batch) * (ctx.world if ctx is not None else 1)
    outfile = None
    try:
        outfile = open(my_output, "w") if my_output else None       # inside the try: a bad path must not leak the scratch FASTA
        reference_seqs = dict(zip(*parse_fasta(reference_db_path, return_names=True)))
        # The reference calls generate_single once per template and requested sequence, in this order, and prints each sequence
        # as soon as it exists (/root/reference/src/pgen/pgen_msa_revised.py:107-115, flush=True).  Here the (template, i) jobs are
        # collected in that order and handed to generate_single_batch a CHUNK at a time: the interpreter RNG (one shuffle per
        # MSA and pass, MSA-major) and the torch draw seeds (one per MSA, in order) are consumed exactly as in the serial loop,
        # so the strings are the same for any chunk size; equal-shape alignments of a chunk share forwards; with
        # sampler.shard_over_ranks a chunk is split over the torch.distributed ranks.  After every chunk the finished records
        # are appended and flushed: a failure on a late template (phmmer, mafft, out of memory) loses that chunk only.
        pending = []                                    # (name, alignment, excluded columns)
        row = -1 if legacy else 0

        def flush_pending():
            if not pending:
                return
            new_seqs = sampler.generate_single_batch([p[1] for p in pending], steps=steps, passes=passes, burn_in=burn_in, k=top_k,
                                                     target_index=row, exclude_positions=[p[2] for p in pending],
                                                     max_batch=template_batch)
            if outfile is not None:
                for (name, _, _), new_seq in zip(pending, new_seqs):
                    print(f">{name}\n{new_seq.replace('-', '')}", file=outfile, flush=True)
            del pending[:]

        for template_name, template_seq in templates:
            alignment, exclude_positions, row = build_template_alignment(
                template_name, template_seq, reference_seqs, reference_db_path, alignment_size, keep_identical, ep, op,
                legacy, gap_percent_threshold, debug)
            for i in range(seqs_per_template):
                pending.append((f"{i}_{template_name}", alignment, exclude_positions))
                if len(pending) >= chunk_jobs:
                    flush_pending()
        flush_pending()
    finally:
        if outfile is not None:
            outfile.close()
        os.unlink(reference_db_path)


def build_parser():
    parser = argparse.ArgumentParser(description=textwrap.dedent("""Samples from the ESM-MSA model to generate new protein sequences."""),
                                     formatter_class=RawAndDefaultsFormatter)
    parser.add_argument("--templates", default=None, required=True, help="an unaligned fasta file with sequences to mask for generating new sequences.")
    parser.add_argument("--references", default=None, required=True, help="an unaligned fasta file with reference sequences to search for homologs to the templates.")
    parser.add_argument("-o", default=None, required=True, help="a fasta file to write generated sequences to")
    parser.add_argument("--seqs_per_template", type=int, default=1, help="Number of new sequences to generate for each template sequence.")
    parser.add_argument("--keep_identical", action="store_true", default=False, help="keep reference sequences identical to the template (thrown out by default).")
    parser.add_argument("--steps", type=int, default=10, help="Randomly assign the input positions to this many mask bins, and mask and generate over one bin at a time.")
    parser.add_argument("--passes", type=int, default=3, help="how many passes over the entire template sequence to make.")
    parser.add_argument("--burn_in", type=int, default=1, help="this many passes sample from the entire distribution, afterwards from the top_k most likely.")
    parser.add_argument("--top_k", type=int, default=1, help="Sample from this many of the most probable amino acids after burn in. 0 = always the full distribution.")
    parser.add_argument("--legacy", action="store_true", default=False, help="sample the last sequence of the MSA rather than the first and ignore gap_percent_threshold.")
    parser.add_argument("--gap_percent_threshold", type=float, default=80.0,
                        help="Don't resample positions where more than this percent of sequences in the alignment contain gaps. Ignored in legacy mode.")
    parser.add_argument("--ep", type=float, default=0.0, help="ep parameter passed to MAFFT for alignments")
    parser.add_argument("--op", type=float, default=1.53, help="op parameter passed to MAFFT for alignments")
    parser.add_argument("--device", type=str, default="gpu", help="gpu (cuda:0) or cuda:[int]")
    parser.add_argument("--model", type=str, default="esm_msa1", choices=sorted(model_map), help="which model to use")
    parser.add_argument("--alignment_size", type=int, default=32, help="how many sequences (template plus references) should be in the alignments used for sequence generation.")
    parser.add_argument("--debug", action="store_true", default=False, help="run phmmer in --max mode (no pre-filters; finds very short hits).")
    parser.add_argument("--template_batch", type=int, default=4, help="resample up to this many equal-shape alignments per forward pass (does not change the output).")
    parser.add_argument("--shard_templates", action="store_true", default=False,
                        help="under torchrun (one process per GPU): split the list of templates over the ranks (contiguous blocks, "
                             "one RCCL gather of the resampled rows at the end); the output is identical to a single-GPU run.")
    add_engine_args(parser)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    seed_everything(args.seed)
    if args.shard_templates and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        args.device = "cuda:%d" % local
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))       # "nccl" is RCCL on ROCm
    sampler = ESM_MSA_sampler(model_map[args.model](checkpoint=args.checkpoint, precision=args.precision, synthetic=args.synthetic_weights), device=args.device)
    sampler.shard_over_ranks = args.shard_templates
    pgen_msa(args.templates, args.references, args.o, args.seqs_per_template, args.keep_identical, args.steps, args.passes,
             args.burn_in, args.device, args.model, args.alignment_size, args.ep, args.op, args.top_k, legacy=args.legacy,
             gap_percent_threshold=args.gap_percent_threshold, debug=args.debug, sampler=sampler, template_batch=args.template_batch)


if __name__ == "__main__":
    main(sys.argv[1:])
