"""Multi-GPU sharding of independent chains: one process per GPU, contiguous blocks of chains,
no data-path collective, ONE gather of the final int token buffers.

Results are independent of the number of ranks because (a) every rank derives its slice of the target-position
table from the same CPython-exact stream (the whole table is generated natively on every rank: microseconds), and
(b) token draws are keyed by the *global* chain id (`pg_sample_params.row_id_base`).
"""
import numpy as np

from . import pyrandom


def shard_range(n_items, world_size, rank):
    """Contiguous block [lo, hi) of rank `rank`; the first n_items % world_size ranks get one more."""
    q, r = divmod(n_items, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def global_position_table(rng, population, P, n_iters, n_rows_total):
    """[n_iters][n_rows_total][P] int32 from ONE stream: iteration-major, row-minor, exactly the order in which the
    reference would call random.sample (esm_sampler.py:242-246)."""
    return rng.sample_table(population, P, n_iters * n_rows_total).reshape(n_iters, n_rows_total, P)


def local_slice(table, lo, hi):
    return np.ascontiguousarray(table[:, lo:hi])


def gather_tokens(dist, local_tokens, counts=None):
    """The one collective: all-gather of the final token buffers (equal shards -> all_gather_into_tensor,
    ragged -> all_gather of padded blocks).  `dist` is torch.distributed; works on RCCL ("nccl") and gloo."""
    import torch
    world = dist.get_world_size()
    if counts is None or len(set(counts)) == 1:
        out = torch.empty((world * local_tokens.shape[0],) + tuple(local_tokens.shape[1:]), dtype=local_tokens.dtype,
                          device=local_tokens.device)
        if dist.get_backend() == "gloo":
            parts = list(out.chunk(world))
            dist.all_gather(parts, local_tokens.contiguous())
        else:
            dist.all_gather_into_tensor(out, local_tokens.contiguous())
        return out
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(local_tokens.shape[1:]), dtype=local_tokens.dtype, device=local_tokens.device)
    pad[:local_tokens.shape[0]] = local_tokens
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:c] for p, c in zip(parts, counts)])
