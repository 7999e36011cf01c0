"""Multi-GPU sharding of independent chains: one process per GPU, contiguous blocks of chains,
no data-path collective, ONE gather of the final int token buffers.

Results are independent of the number of ranks because (a) every rank derives its slice of the target-position
table from the same CPython-exact stream (the whole table is generated natively on every rank: microseconds), and
(b) token draws are keyed by the *global* chain id (`pg_sample_params.row_id_base`).
"""
import random

import numpy as np

from . import pyrandom


class DistContext:
    """torch.distributed as the samplers see it: rank, world size and the module itself."""

    def __init__(self, dist):
        self.dist, self.rank, self.world = dist, dist.get_rank(), dist.get_world_size()


def dist_context():
    """A DistContext when this process is one of several torch.distributed ranks (RCCL or gloo), else None.
    The samplers then split every batch of chains / MSAs contiguously over the ranks."""
    try:
        import torch.distributed as dist
    except ImportError:
        return None
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return None
    return DistContext(dist)


def broadcast_object(ctx, obj):
    """rank 0's `obj` on every rank (host-side control data: seeds, RNG states)."""
    box = [obj]
    ctx.dist.broadcast_object_list(box, src=0)
    return box[0]


def sharding_requested(flag):
    """Batch sharding over torch.distributed ranks is OPT-IN: `sampler.shard_over_ranks = True` or PGIBBS_SHARD_OVER_RANKS=1.
    The natural way to data-parallelise the reference API under torchrun is every rank sampling its OWN seeds; a sampler that
    silently split each rank's batch and gathered rows across ranks would mix different jobs."""
    import os
    return bool(flag) or os.environ.get("PGIBBS_SHARD_OVER_RANKS", "0") not in ("", "0")


def job_digest(*parts):
    """Stable digest of the arguments that define a sharded job (every rank must have been called with the same ones)."""
    import hashlib
    h = hashlib.sha256()
    for p in parts:
        h.update(repr(p).encode())
        h.update(b"\0")
    return h.hexdigest()


def check_same_job(ctx, digest, what):
    """Raise on EVERY rank (no hang in the later collective) unless all ranks passed the same job to `what`."""
    got = [None] * ctx.world
    ctx.dist.all_gather_object(got, digest)
    if len(set(got)) != 1:
        raise ValueError("%s with shard_over_ranks: the ranks were called with different arguments (seed sequences, batch_size, "
                         "indexes, num_iters, ...); sharding splits ONE job over the ranks -- to run independent jobs per rank "
                         "leave shard_over_ranks off (digests per rank: %s)" % (what, [g[:8] for g in got]))


def sync_host_rng(ctx):
    """Every rank continues from rank 0's interpreter RNG state, so all ranks derive the same position tables (each rank
    generates the whole table natively and slices its block: no per-iteration communication)."""
    random.setstate(broadcast_object(ctx, random.getstate()))


def run_sharded(ctx, tokens, table, row_id_base, rows_per_item, run_fn, device=None, guard=None):
    """One batch over the ranks: item block [lo, hi) of `tokens` [B, ...] and `table` [iters, B, ...] goes through
    run_fn(local_tokens, local_table, local_row_id_base) (in place), then ONE all-gather rebuilds the whole token buffer
    on every rank.  rows_per_item = Philox row ids consumed per item (1 per chain, R per MSA).
    guard: the NativeMaskedLM behind run_fn.  With precision="auto" on fp16 operands a block whose logits leave the fp16 range is
    run again in bf16 (engine.py); the single-GPU call would then have run the WHOLE batch in bf16, so the ranks agree first (one
    all-gather of a flag) and, if any block overflowed, every other rank repeats its block in bf16 too -- the gathered result
    stays bit-identical with the single-GPU run."""
    import torch
    B = tokens.shape[0]
    lo, hi = shard_range(B, ctx.world, ctx.rank)
    local = np.ascontiguousarray(tokens[lo:hi])
    agree = guard is not None and bool(getattr(guard, "auto_fp16", False))
    if agree:
        guard.take_fell_back()
    if hi > lo:
        run_fn(local, np.ascontiguousarray(table[:, lo:hi]), row_id_base + lo * rows_per_item)
    if agree:
        mine = guard.take_fell_back()
        flags = [None] * ctx.world
        ctx.dist.all_gather_object(flags, bool(mine))
        if any(flags) and not mine and hi > lo:
            local = np.ascontiguousarray(tokens[lo:hi])
            with guard.forced_bf16():
                run_fn(local, np.ascontiguousarray(table[:, lo:hi]), row_id_base + lo * rows_per_item)
    counts = [shard_range(B, ctx.world, r)[1] - shard_range(B, ctx.world, r)[0] for r in range(ctx.world)]
    t = torch.from_numpy(local)
    on_gpu = ctx.dist.get_backend() != "gloo"
    if on_gpu:
        t = t.to(device if device is not None else "cuda")
    full = gather_tokens(ctx.dist, t, counts)
    return full.cpu().numpy() if on_gpu else full.numpy()


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


def gather_tokens(dist, local_tokens, counts=None, force_padded=False):
    """The one collective: all-gather of the final token buffers (equal shards -> all_gather_into_tensor,
    ragged -> all_gather of padded blocks; force_padded takes the ragged form for equal shards too -- how the single-GPU
    RCCL test executes it).  `dist` is torch.distributed; works on RCCL ("nccl") and gloo."""
    import torch
    world = dist.get_world_size()
    if counts is None or (len(set(counts)) == 1 and not force_padded):
        out = torch.empty((world * local_tokens.shape[0],) + tuple(local_tokens.shape[1:]), dtype=local_tokens.dtype,
                          device=local_tokens.device)
        if dist.get_backend() == "gloo":
            parts = list(out.chunk(world))
            dist.all_gather(parts, local_tokens.contiguous())
        else:
            dist.all_gather_into_tensor(out, local_tokens.contiguous())
        return out
    mx = max(counts) + (1 if force_padded else 0)
    pad = torch.zeros((mx,) + tuple(local_tokens.shape[1:]), dtype=local_tokens.dtype, device=local_tokens.device)
    pad[:local_tokens.shape[0]] = local_tokens
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:c] for p, c in zip(parts, counts)])


class NativeComm:
    """The same collective behind the C ABI (include/pgibbs.h: pg_comm_*, pg_gather_tokens -- RCCL opened by the library itself,
    no torch.distributed in the data path).  The communicator id is the only thing that travels out of band: rank 0 makes it,
    `exchange(id_bytes_or_None) -> id_bytes` hands it to the others (default: a broadcast over an existing torch.distributed
    group of any backend; a world of one needs no exchange)."""

    def __init__(self, rank, world, device_ordinal, exchange=None):
        import ctypes
        from . import _lib
        self._lib = _lib
        L = _lib.lib()
        buf = ctypes.create_string_buffer(_lib.PG_COMM_ID_BYTES)
        if rank == 0:
            _lib.check(L.pg_comm_unique_id(buf))
        if world > 1:
            if exchange is None:
                import torch.distributed as dist
                box = [buf.raw if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                raw = box[0]
            else:
                raw = exchange(buf.raw if rank == 0 else None)
            buf = ctypes.create_string_buffer(raw, _lib.PG_COMM_ID_BYTES)
        self.rank, self.world, self.device = rank, world, device_ordinal
        self.handle = ctypes.c_void_p()
        _lib.check(L.pg_comm_create(rank, world, buf, device_ordinal, ctypes.byref(self.handle)))

    def gather_tokens(self, local_tokens, counts=None, stream=None):
        """local_tokens: int32 CUDA tensor [rows, width...] on this communicator's device -> [sum(counts), width...] (a torch
        tensor only as the owner of device memory; the call takes raw pointers).  Runs on `stream` (default: torch's current
        stream) and returns after enqueueing."""
        import ctypes
        import torch
        t = local_tokens.contiguous()
        assert t.dtype == torch.int32 and t.is_cuda and t.device.index == self.device
        rows = t.shape[0]
        width = int(np.prod(t.shape[1:])) if t.dim() > 1 else 1
        counts = [rows] * self.world if counts is None else [int(c) for c in counts]
        out = torch.empty((sum(counts),) + tuple(t.shape[1:]), dtype=torch.int32, device=t.device)
        c_counts = (ctypes.c_int64 * self.world)(*counts)
        s = stream if stream is not None else torch.cuda.current_stream(t.device).cuda_stream
        self._lib.check(self._lib.lib().pg_gather_tokens(self.handle, ctypes.c_void_p(s), ctypes.c_void_p(t.data_ptr()), rows, width,
                                                         c_counts, ctypes.c_void_p(out.data_ptr())))
        return out

    def close(self):
        if self.handle:
            self._lib.lib().pg_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
