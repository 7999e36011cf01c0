"""MI355X-native Gibbs-sampling engine for masked protein language models.

Drop-in for the hot path of seanrjohnson/protein_gibbs_sampler (`pgen`):
    from protein_gibbs_sampler_amd import models, esm_sampler, esm_msa_sampler
    sampler = esm_sampler.ESM_sampler(models.ESM1b(), device="gpu")
    sampler.generate(n_samples=256, seed_seq=..., batch_size=256, num_iters=50, num_positions_percent=10)

All per-iteration work (mask scatter, ESM-1b / ESM-MSA-1b forward, top-k/temperature categorical draw,
token write-back) runs in hand-written gfx950 kernels behind the C ABI of include/pgibbs.h.  There is
no CPU fallback: without the built library or without a GPU the compute entry points raise.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
__all__ = ["models", "esm_sampler", "esm_msa_sampler", "alphabet", "weights", "engine", "pyrandom"]
