"""Drop-in for `pgen.esm_sampler` (/root/reference/src/pgen/esm_sampler.py): same class, method names,
arguments, defaults, exceptions and messages; the per-iteration work runs on an MI355X.

What is kept verbatim in behaviour (citations = reference lines):
  * device grammar "cpu" | "gpu" | "cuda:N" and its three error messages            :66-78
  * seed cleaning / "<mask>" padding / list seeds drawn with random.choices           :95-126
  * num_positions / leader_length derivation and clamps, `indexes` rebinding (Q1)     :186-207
  * consumption order of the interpreter's RNG (choices per batch, sample per chain)  :112, :242-246
  * last-batch truncation, untokenize with "<mask>" left as text (Q6)                  :84-93, :236-239
What changes: the Python loops of :209-234 become one native call per batch
(`NativeMaskedLM.gibbs_run`), or -- for plug-in models that are not the HIP engine -- a device loop
whose mask scatter and draw are HIP kernels (`_gibbs.run_plugin_loop`).  The token draw uses the
engine's counter-based generator (pg_draw v1) keyed from torch's global RNG, since torch's serial
CPU multinomial stream cannot be reproduced by a data-parallel kernel (DESIGN.md "RNG contract").
"""
import ctypes
import math
import random
import re

import numpy as np
import torch
from tqdm import trange

from . import _gibbs, _lib, sharding
from .engine import NativeMaskedLM

ESM_ALLOWED_AMINO_ACIDS = "ACDEFGHIKLMNPQRSTVWY"

_step_counter = [0]


def _device_of(t):
    return t.device if isinstance(t, torch.Tensor) else torch.device("cpu")


def generate_step(out, gen_idx, temperature=None, top_k=0, sample=False, valid_idx=None, rng_seed=None, counter=None):
    """Generate a token id from out[gen_idx] (reference :8-45) on the GPU.

    out: logits [seq_len, vocab] (tensor or array); returns a 0-d int64 tensor like the reference.
    The draw is the HIP kernel behind pg_sample_writeback_device with P = 1.  `rng_seed`/`counter`
    pin the draw; by default a fresh counter value is used per call and the key comes from torch's
    global generator, so repeated calls are independent draws as in the reference.
    """
    if not torch.cuda.is_available():
        raise RuntimeError("generate_step needs an MI355X: the draw is a HIP kernel, there is no CPU implementation")
    dev = out.device if isinstance(out, torch.Tensor) and out.device.type == "cuda" else torch.device("cuda:0")
    logits = torch.as_tensor(out, dtype=torch.float32).to(dev).contiguous()
    if logits.dim() != 2:
        raise ValueError("out must be [seq_len, vocab]")
    width, V = logits.shape
    if valid_idx is None:
        valid_idx = list(range(V))
    if len(valid_idx) > 32:
        raise ValueError("at most 32 valid tokens are supported")
    if rng_seed is None:
        rng_seed = int(torch.randint(0, 2**62, (1,)).item())
    if counter is None:
        counter = _step_counter[0]
        _step_counter[0] += 1
    params = _lib.make_sample_params(False, 0, top_k, float("inf") if sample else 0, temperature, list(valid_idx), rng_seed,
                                     rng_stream=counter >> 32, row_id_base=counter & 0xFFFFFFFF)
    tok = torch.zeros((1, width), dtype=torch.int32, device=dev)
    idx = torch.tensor([[int(gen_idx) % width]], dtype=torch.int32, device=dev)
    picked = torch.empty((1, 1), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(_lib.lib().pg_sample_writeback_device(stream, ctypes.c_void_p(tok.data_ptr()), 1, width,
                                                         ctypes.c_void_p(logits.data_ptr()), V, ctypes.c_void_p(idx.data_ptr()),
                                                         None, 1, 1, ctypes.byref(params), 0,
                                                         ctypes.c_void_p(picked.data_ptr())))
    return torch.tensor(int(picked.item()))


class ESM_sampler():
    """adapted from bert-gen bert-babble.ipynb (via pgen.esm_sampler.ESM_sampler)"""

    def __init__(self, model, device="cpu"):
        """model: an object with attributes model, alphabet and batch_converter (reference :54-58)."""
        self.model = model
        self.model.model = self.model.model.eval()
        self.device, self.cuda = _gibbs.resolve_device(device)
        self.model.model.to(self.device)
        self.valid_aa_idx = sorted([self.model.alphabet.get_idx(tok) for tok in ESM_ALLOWED_AMINO_ACIDS])
        # key of the token-draw generator; None -> drawn from torch's global RNG once per generate()
        self.draw_seed = None
        self.rng_stream = 0
        # filled by generate(..., ) when self.record is True: per-batch dicts with tables / sampled logits / tokens
        self.record = False
        self.last_run = []
        # when torch.distributed is initialised with several ranks, generate() shards each batch over them
        self.shard_over_ranks = False    # opt-in (or PGIBBS_SHARD_OVER_RANKS=1): generate() splits every batch of ONE job over the torch.distributed ranks

    # ---- helpers with the reference's names ----------------------------------------------------
    def untokenize_batch(self, batch, bos, eos):
        start_offset = 1 if bos else 0
        end_offset = -1 if eos else 0
        if hasattr(batch, "numpy") and getattr(batch, "ndim", 0) == 2:       # a [B, T] token tensor: one table lookup per row
            arr = batch.numpy()
            return self.model.alphabet.decode_rows(arr[:, start_offset:arr.shape[1] + end_offset])
        if hasattr(batch, "tolist"):
            batch = batch.tolist()
        return ["".join([self.model.alphabet.get_tok(seq[i]) for i in range(0 + start_offset, len(seq) + end_offset)])
                for seq in batch]

    @staticmethod
    def clean_seed_seq(seed_to_clean):
        return _gibbs.clean_seed(seed_to_clean, ESM_ALLOWED_AMINO_ACIDS)

    def get_init_seq(self, seed_seq, max_len, batch_size=1):
        """Initial token batch: seeds right-padded with <mask> (reference :104-126).  A list of seeds consumes the
        interpreter's RNG once per batch (random.choices, :112) BEFORE any position draw."""
        if isinstance(seed_seq, list):
            seeds = random.choices(seed_seq, k=batch_size)
        elif isinstance(seed_seq, str):
            seeds = [seed_seq] * batch_size
        else:
            raise Exception("seed sequence should either be a string or list")
        rows = [(str(i), _gibbs.mask_padded(seed, max_len, ESM_ALLOWED_AMINO_ACIDS)) for i, seed in enumerate(seeds)]
        return self.model.batch_converter(rows)[2]

    def get_random_target_index(self, batch_size, indexes, num_positions):
        """== [random.sample(indexes, num_positions) for b in range(batch_size)] (reference :242-246),
        produced by the native CPython-exact generator on the interpreter's global RNG state."""
        from . import pyrandom
        return pyrandom.global_sample_table(list(indexes), num_positions, batch_size).tolist()

    def get_target_index_in_order(self, batch_size, indexes, next_i, num_positions):
        last_i, target = _gibbs.in_order_window(indexes, next_i, num_positions)
        return last_i, [target] * batch_size

    def mask_target_indexes(self, batch, target_indexes):
        """Reference :259-262.  Nested lists (as in the reference's unit test) are masked on the host; a
        device token tensor goes through the HIP scatter kernel."""
        mask_idx = self.model.alphabet.mask_idx
        if isinstance(batch, torch.Tensor) and batch.device.type == "cuda":
            P = max((len(t) for t in target_indexes), default=0)
            table = np.full((len(target_indexes), P), -1, dtype=np.int32)
            for b, t in enumerate(target_indexes):
                table[b, :len(t)] = t
            tok = batch.to(torch.int32).contiguous()
            d_table = torch.from_numpy(table).to(batch.device)
            with torch.cuda.device(batch.device):
                stream = ctypes.c_void_p(torch.cuda.current_stream(batch.device).cuda_stream)
                _lib.check(_lib.lib().pg_mask_scatter_device(stream, ctypes.c_void_p(tok.data_ptr()), tok.shape[0], tok.shape[1],
                                                             ctypes.c_void_p(d_table.data_ptr()), None, len(target_indexes), P,
                                                             mask_idx))
            batch.copy_(tok.to(batch.dtype))
            return
        for batch_index in range(len(target_indexes)):
            for kk in target_indexes[batch_index]:
                batch[batch_index][kk] = mask_idx

    def calculate_indexes(self, indexes, leader_length, max_len, rollover_from_start):
        return _gibbs.candidate_indexes(indexes, leader_length, max_len, rollover_from_start)

    # ---- the sampler -----------------------------------------------------------------------------
    def generate(self, n_samples, seed_seq, batch_size=1, in_order=False, max_len=None, leader_length=0,
                 leader_length_percent=None, top_k=0, temperature=None, num_iters=10, burnin=float('inf'), mask=True,
                 num_positions=0, num_positions_percent=None, indexes=None, rollover_from_start=False,
                 show_progress_bar=True):
        """generate sequences -- arguments exactly as pgen.esm_sampler.ESM_sampler.generate (reference :128-170)."""
        if isinstance(seed_seq, str):
            sequence_length = len(seed_seq)
        elif isinstance(seed_seq, list):
            sequence_length = max(len(seed) for seed in seed_seq)
        else:
            raise ValueError("Unknown seed sequence format, expecting str or list")

        sequences = []
        n_batches = math.ceil(n_samples / batch_size)
        if max_len is None:
            max_len = sequence_length
        num_positions, leader_length = _gibbs.derive_counts(max_len, num_positions, num_positions_percent, leader_length,
                                                            leader_length_percent)

        if not self.cuda:
            raise RuntimeError("ESM_sampler.generate needs device 'gpu'/'cuda:N' on an MI355X: this package implements the "
                               "Gibbs hot path as HIP kernels only and has no CPU implementation")
        draw_seed = self.draw_seed if self.draw_seed is not None else int(torch.randint(0, 2**62, (1,)).item())
        native = isinstance(self.model.model, NativeMaskedLM)
        self.last_run = []
        # several torch.distributed ranks (one per GPU): every batch is split contiguously over them
        ctx = sharding.dist_context() if (native and sharding.sharding_requested(self.shard_over_ranks)) else None
        if ctx is not None:
            if self.record:
                raise ValueError("record=True is not supported together with shard_over_ranks (per-draw logits stay on their rank)")
            sharding.check_same_job(ctx, sharding.job_digest(
                n_samples, seed_seq, batch_size, in_order, max_len, leader_length, top_k, temperature, num_iters, burnin, mask,
                num_positions, None if indexes is None else list(indexes), rollover_from_start, self.rng_stream), "ESM_sampler.generate")
            sharding.sync_host_rng(ctx)
            draw_seed = sharding.broadcast_object(ctx, draw_seed)

        for batch_n in trange(n_batches, disable=(not show_progress_bar)):
            batch = self.get_init_seq(seed_seq, max_len, batch_size)

            indexes, last_i = self.calculate_indexes(indexes, leader_length, max_len, rollover_from_start)
            indexes = _gibbs.normalise_indexes(indexes, batch.shape[1])     # IndexError / negative wrap as batch[b][kk]
            if num_positions > len(indexes):
                num_positions = len(indexes)

            table, last_i = _gibbs.build_target_table(num_iters, (batch_size,), indexes, num_positions, in_order, last_i)
            params = _lib.make_sample_params(mask, self.model.alphabet.mask_idx, top_k, burnin, temperature, self.valid_aa_idx,
                                             draw_seed, rng_stream=self.rng_stream, row_id_base=batch_n * batch_size)
            if native and ctx is not None:
                def run_block(ltok, ltable, base):
                    params.row_id_base = base & 0xFFFFFFFF
                    self.model.model.set_job_items(batch.shape[0])      # shard of a batch.shape[0]-item job
                    try:
                        self.model.model.gibbs_run(ltok, ltable, params)
                    finally:
                        self.model.model.set_job_items(0)
                tok = sharding.run_sharded(ctx, np.ascontiguousarray(batch.numpy(), dtype=np.int32), table,
                                           batch_n * batch_size, 1, run_block, self.device, guard=self.model.model)
                batch = torch.from_numpy(tok.astype(np.int64))
            elif native:
                tok = np.ascontiguousarray(batch.numpy(), dtype=np.int32)
                lg, st = self.model.model.gibbs_run(tok, table, params, want_logits=self.record, want_tokens=self.record)
                batch = torch.from_numpy(tok.astype(np.int64))
                if self.record:
                    self.last_run.append(dict(table=table, sampled_logits=lg, sampled_tokens=st, tokens=tok.copy()))
            else:
                batch = _gibbs.run_plugin_loop(self.model.model, batch, table, params, self.device)
                if self.record:
                    self.last_run.append(dict(table=table, tokens=batch.numpy().copy()))

            strs = self.untokenize_batch(batch, self.model.alphabet.prepend_bos, self.model.alphabet.append_eos)
            if batch_n == (n_batches - 1):
                sequences += strs[0:n_samples - len(sequences)]
            else:
                sequences += strs
        return sequences

    # ---- masked log-likelihood (reference :277-363) ---------------------------------------------------
    def log_likelihood(self, seq, with_masking=True, verbose=False, mask_distance=float("inf"), batch_size=None):
        """(mean log-likelihood, per-position list) of one sequence -- see log_likelihood_batch."""
        return next(self.log_likelihood_batch([seq], with_masking, verbose, mask_distance, batch_size))

    def log_likelihood_batch(self, seq_list, with_masking=True, verbose=False, mask_distance=float("inf"), batch_size=None):
        """Same contract as pgen.esm_sampler.ESM_sampler.log_likelihood_batch: yields (float mean, list[float]).
        with_masking: min(mask_distance, len) copies of the sequence, copy i masked at every
        num_copies-th position starting at i; the log-probability of the original residue is read at the masked
        positions.  The forward, log-softmax and gather run on the GPU (pg_esm_forward_logprobs: the LM head is
        evaluated only at the scored rows)."""
        if not self.cuda:
            raise RuntimeError("ESM_sampler.log_likelihood_batch needs device 'gpu'/'cuda:N' on an MI355X: "
                               "there is no CPU implementation")
        n_batches = len(seq_list)
        if batch_size is None:
            batch_size = n_batches
        reformatted_seq = [(str(idx), self.clean_seed_seq(seq)) for idx, seq in enumerate(seq_list)]
        _, _, tokens = self.model.batch_converter(reformatted_seq)
        range_start = 1 if self.model.alphabet.prepend_bos else 0
        end_modifier = -1 if self.model.alphabet.append_eos else 0
        batch_range_end = [len(seq) + range_start for seq in seq_list]
        overall_range_end = tokens.shape[1] + end_modifier
        assert max(len(seq) for seq in seq_list) == len(range(range_start, overall_range_end))
        old_toks = tokens.numpy()
        mask_idx = self.model.alphabet.mask_idx
        for seq_idx in range(len(reformatted_seq)):
            original_string = reformatted_seq[seq_idx][1]
            end = batch_range_end[seq_idx]
            # tokens of THIS sequence alone (no padding reaches the model, as in the reference's masked path)
            _, _, one = self.model.batch_converter([(0, original_string)])
            if with_masking:
                n = int(min(mask_distance, len(original_string)))
                copies = one.repeat(n, 1)
                pos_of = [list(range(range_start + i, end, n)) for i in range(n)]
                for i, pos in enumerate(pos_of):
                    copies[i, pos] = mask_idx
                assert sum(len(p) for p in pos_of) == len(original_string)
            else:
                n = 1
                copies = one
                pos_of = [list(range(range_start, end))]
            P = max((len(p) for p in pos_of), default=0)
            idx = np.full((n, P), -1, dtype=np.int32)
            tgt = np.zeros((n, P), dtype=np.int32)
            for i, pos in enumerate(pos_of):
                idx[i, :len(pos)] = pos
                tgt[i, :len(pos)] = old_toks[seq_idx, pos]
            likelihood_sum = np.float32(0.0)
            likelihood_list = []
            for batch_start in range(0, n, max(1, batch_size)):
                sl = slice(batch_start, batch_start + max(1, batch_size))
                nb = copies[sl].shape[0]
                lp = _gibbs.score_positions(self.model.model, copies[sl], np.arange(nb), idx[sl], tgt[sl], self.device)
                for i in range(nb):
                    for p in range(len(pos_of[batch_start + i])):
                        likelihood_sum = np.float32(likelihood_sum + lp[i, p])
                        likelihood_list.append(float(lp[i, p]))
            if with_masking:
                # the reference appends in (copy, position) order; positions of one copy ascend (stride n)
                pass
            yield (float(likelihood_sum / np.float32(len(seq_list[seq_idx]))), likelihood_list)
