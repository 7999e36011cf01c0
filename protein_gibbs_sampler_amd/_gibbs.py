"""Host-side pieces shared by ESM_sampler and ESM_MSA_sampler: target-position tables and the
device loop for plug-in (non-engine) models.  Everything that touches tokens or logits per position
is a HIP kernel behind the C ABI; this file only prepares index tables and sequences the calls.
"""
import ctypes

import numpy as np

from . import _lib
from . import pyrandom as _pyr

SHADOW_BIT = 1 << 30   # include/pgibbs.h: sampled but not written (a later duplicate in the same row wins)


def resolve_device(device):
    """The reference's device grammar (esm_sampler.py:66-78, esm_msa_sampler.py:47-61): "cpu" | "gpu" | "cuda:N", with
    its three error messages.  Returns (device string, uses_gpu)."""
    import re

    import torch
    if device == "gpu":
        device = "cuda:0"
    if re.match("^cuda:[0-9]+$", device):
        if not torch.cuda.is_available():
            raise Exception("gpu requested, but No Cuda devices found")
        if int(device.split(":")[1]) >= torch.cuda.device_count():
            raise Exception("Invalid cuda device number: " + device)
        return device, True
    if device != "cpu":
        raise Exception("Invalid device: " + device)
    return device, False


def clean_seed(seq, allowed):
    """clean_seed_seq of both samplers (esm_sampler.py:95-102, esm_msa_sampler.py:93-99)."""
    seq = seq.upper()
    bad = set(seq) - set(allowed)
    if bad:
        raise Exception("Invalid input character: " + ",".join(bad))
    return seq


def mask_padded(seq, max_len, allowed):
    """A cleaned seed right-padded with the literal "<mask>" up to max_len residues (esm_sampler.py:115,120)."""
    return clean_seed(seq, allowed) + "<mask>" * (max_len - len(seq))


def candidate_indexes(indexes, leader_length, max_len, rollover_from_start):
    """calculate_indexes (esm_sampler.py:264-274, esm_msa_sampler.py:294-304): 1-based token positions after the leader
    (position 0 is <cls>), or the caller's `indexes` untouched; last_i = where an in-order sweep starts."""
    if indexes is not None:
        return indexes, -1
    indexes = range(1, max_len + 1)
    if rollover_from_start:
        return indexes, -1
    return indexes[leader_length:], leader_length - 1


def derive_counts(length, num_positions, num_positions_percent, leader_length, leader_length_percent):
    """num_positions / leader_length from their *_percent forms, clamped at 0 (esm_sampler.py:189-197)."""
    if num_positions_percent is not None:
        num_positions = int(length * (num_positions_percent / 100))
    if leader_length_percent is not None:
        leader_length = int(length * (leader_length_percent / 100))
    return max(num_positions, 0), max(leader_length, 0)


def in_order_window(indexes, next_i, num_positions):
    """get_target_index_in_order (/root/reference/src/pgen/esm_sampler.py:248-257)."""
    out = []
    n = len(indexes)
    for _ in range(num_positions):
        next_i = (next_i + 1) % n
        out.append(indexes[next_i])
    return next_i, out


def normalise_indexes(indexes, width):
    """Candidate positions as the reference's `batch[b][kk]` would resolve them (esm_sampler.py:234,262): a torch row of
    `width` tokens accepts -width <= kk < width, negative values counting from the end; anything else is the IndexError
    torch raises there.  `range` objects (the default candidates) are returned untouched."""
    if isinstance(indexes, range):
        if len(indexes) and (indexes[0] < -width or indexes[-1] >= width or indexes[0] < 0):
            indexes = list(indexes)
        else:
            return indexes
    out = []
    for kk in indexes:
        k = int(kk)
        if k < -width or k >= width:
            raise IndexError("index %d is out of bounds for dimension 0 with size %d" % (k, width))
        out.append(k + width if k < 0 else k)
    return out


def mark_shadowed(table, indexes):
    """Sequential write-back semantics for duplicate positions inside one row's target list."""
    if len(set(indexes)) == len(indexes):
        return table
    flat = table.reshape(-1, table.shape[-1])
    for row in flat:
        seen = set()
        for p in range(len(row) - 1, -1, -1):
            v = int(row[p])
            if v < 0:
                continue
            if v in seen:
                row[p] = v | SHADOW_BIT
            seen.add(v)
    return table


def build_target_table(n_iters, n_rows_shape, indexes, num_positions, in_order, last_i):
    """All target positions of one batch, for every iteration, as int32 [n_iters, *n_rows_shape, P].

    Mirrors the per-iteration branch of generate() (esm_sampler.py:210-218, esm_msa_sampler.py:222-231):
    random -> one random.sample per row in row-major order, iteration-major (exactly the order the
    reference consumes the interpreter's RNG); in order -> one shared cyclic window per iteration;
    num_positions == 0 -> every candidate position every iteration.
    Returns (table, last_i).
    """
    indexes = list(indexes)
    n_rows = int(np.prod(n_rows_shape)) if len(n_rows_shape) else 1
    if num_positions > 0:
        P = num_positions
        if in_order:
            table = np.empty((n_iters, n_rows, P), dtype=np.int32)
            for it in range(n_iters):
                last_i, win = in_order_window(indexes, last_i, P)
                table[it, :, :] = np.asarray(win, dtype=np.int32)[None, :]
        else:
            table = _pyr.global_sample_table(indexes, P, n_iters * n_rows).reshape(n_iters, n_rows, P)
    else:
        P = len(indexes)
        table = np.broadcast_to(np.asarray(indexes, dtype=np.int32), (n_iters, n_rows, P)).copy()
    table = mark_shadowed(table, indexes)
    return table.reshape((n_iters,) + tuple(n_rows_shape) + (P,)), last_i


def _current_stream_ptr(device):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def run_plugin_loop(model_callable, tokens_i64, table, params, device, row_map=None, mask_row_map=None,
                    sample_flags=None):
    """Gibbs loop for a plug-in model whose forward is not the HIP engine (any callable
    tokens[int64, device] -> {"logits": float tensor}): mask scatter and draw/write-back are the HIP kernels
    `pg_mask_scatter_device` / `pg_sample_writeback_device`; the token buffer stays on the device.

    tokens_i64: torch int64 tensor [..rows.., width] (any device); table int32 [n_iters, n_sel, P];
    row_map: optional int32 [n_sel] token-row sampled for each selected row (generate_single);
    mask_row_map: token-row masked for each selected row (defaults to row_map; generate_single masks row -1);
    sample_flags: optional per-iteration override of `sample` (generate_single's pass_num < burn_in).
    Returns the final tokens as a torch int64 CPU tensor of the input shape.
    """
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X visible: the Gibbs hot path has no CPU implementation in this package")
    L = _lib.lib()
    dev = torch.device(device)
    shape = tuple(tokens_i64.shape)
    width = shape[-1]
    n_rows = int(np.prod(shape[:-1]))
    tok = tokens_i64.to(device=dev, dtype=torch.int32).contiguous()
    n_iters, n_sel, P = table.shape
    d_table = torch.from_numpy(np.ascontiguousarray(table)).to(dev)
    d_rowmap = torch.from_numpy(np.ascontiguousarray(row_map, dtype=np.int32)).to(dev) if row_map is not None else None
    rm_ptr = ctypes.c_void_p(d_rowmap.data_ptr()) if d_rowmap is not None else None
    d_mrowmap = (torch.from_numpy(np.ascontiguousarray(mask_row_map, dtype=np.int32)).to(dev)
                 if mask_row_map is not None else None)
    mrm_ptr = ctypes.c_void_p(d_mrowmap.data_ptr()) if d_mrowmap is not None else rm_ptr
    with torch.cuda.device(dev):
        for it in range(n_iters):
            stream = _current_stream_ptr(dev)
            idx_ptr = ctypes.c_void_p(d_table[it].data_ptr())
            if params.mask and P > 0:
                _lib.check(L.pg_mask_scatter_device(stream, ctypes.c_void_p(tok.data_ptr()), n_rows, width, idx_ptr,
                                                    mrm_ptr, n_sel, P, params.mask_idx))
            out = model_callable(tok.to(torch.int64).reshape(shape))["logits"]
            if P == 0:
                continue
            out = out.to(device=dev, dtype=torch.float32).contiguous()
            V = out.shape[-1]
            if sample_flags is not None:
                params.burnin = _lib.INT32_MAX if sample_flags[it] else 0
            _lib.check(L.pg_sample_writeback_device(stream, ctypes.c_void_p(tok.data_ptr()), n_rows, width,
                                                    ctypes.c_void_p(out.data_ptr()), V, idx_ptr, rm_ptr, n_sel, P,
                                                    ctypes.byref(params), it, None))
        torch.cuda.synchronize(dev)
    return tok.to(device="cpu", dtype=torch.int64).reshape(shape)


def score_positions(model_callable, tokens_i64, row_of, idx, targets, device):
    """log_softmax(logits)[target] at (token row row_of[s], position idx[s][p]); idx < 0 -> 0.
    Engine models: one native call (LM head only at the scored rows).  Plug-in models: the caller's forward, then the
    HIP gather kernel `pg_logprob_gather_device` on its device-resident logits."""
    import torch
    from .engine import NativeMaskedLM
    row_of = np.ascontiguousarray(row_of, dtype=np.int32)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    targets = np.ascontiguousarray(targets, dtype=np.int32)
    if isinstance(model_callable, NativeMaskedLM):
        return model_callable.forward_logprobs(tokens_i64.numpy() if hasattr(tokens_i64, "numpy") else tokens_i64, row_of, idx, targets)
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X visible: log-likelihood scoring has no CPU implementation in this package")
    dev = torch.device(device)
    with torch.cuda.device(dev):
        out = model_callable(tokens_i64.to(dev))["logits"].to(device=dev, dtype=torch.float32).contiguous()
        width, V = out.shape[-2], out.shape[-1]
        n_rows = out.numel() // (width * V)
        d_idx, d_row, d_tgt = (torch.from_numpy(a).to(dev) for a in (idx, row_of, targets))
        res = torch.zeros(idx.shape, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().pg_logprob_gather_device(_current_stream_ptr(dev), ctypes.c_void_p(out.data_ptr()), n_rows, width, V,
                                                       ctypes.c_void_p(d_idx.data_ptr()), ctypes.c_void_p(d_row.data_ptr()),
                                                       ctypes.c_void_p(d_tgt.data_ptr()), idx.shape[0], idx.shape[1],
                                                       ctypes.c_void_p(res.data_ptr())))
        torch.cuda.synchronize(dev)
    return res.cpu().numpy()
