"""`pgen.utils` under its own name (/root/reference/src/pgen/utils.py): the reference keeps its FASTA readers, the gap
bookkeeping of `generate_single`'s callers and the mafft / phmmer wrappers in one module, and its callers and tests import them as
`from pgen import utils`.  Here they live in `fasta_io` (readers, writers, SequenceSubsetter: utils.py:87-169, 210-240, 316-363) and
`msa_tools` (unalign / add_gaps_back: utils.py:42-85; add_to_msa, generate_alignment, run_phmmer: utils.py:171-208, 242-314); this
module is the same surface in one place, so that `from protein_gibbs_sampler_amd import utils` is the drop-in spelling.
"""
from .fasta_io import SequenceSubsetter, parse_fasta, parse_fasta_string, write_sequential_fasta
from .msa_tools import (add_gaps_back, add_to_msa, generate_alignment, run_phmmer, unalign,
                        write_partitioned_fasta)

__all__ = ["SequenceSubsetter", "parse_fasta", "parse_fasta_string", "write_sequential_fasta", "write_partitioned_fasta",
           "unalign", "add_gaps_back", "add_to_msa", "generate_alignment", "run_phmmer"]
