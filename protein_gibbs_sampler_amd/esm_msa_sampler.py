"""Drop-in for `pgen.esm_msa_sampler` (/root/reference/src/pgen/esm_msa_sampler.py): same class, method
names, arguments, defaults and errors; per-iteration work on an MI355X.

Kept in behaviour (citations = reference lines): `partition` :13-31; device grammar :47-61; seed
cleaning with the 21-symbol alphabet :93-99; `<mask>` padding through the patched MSA batch converter
:78-90; `generate` bookkeeping (rounds, num_positions from sequence_length, `indexes` rebinding)
:189-218; RNG consumption order (one random.sample per (msa, row), :272-279; one random.shuffle per
pass, :129); `generate_single` masking row -1 while sampling `target_index` (quirk Q2, :133-145).
"""
import math
import random
import re

import numpy as np
import torch
from tqdm import trange

from . import _gibbs, _lib, sharding
from .engine import NativeMaskedLM
from .esm_sampler import generate_step  # noqa: F401  (same re-export as the reference, :6)

ESM_MSA_ALLOWED_AMINO_ACIDS = "-ACDEFGHIKLMNPQRSTVWY"
ESM_MSA_GAP_CHARACTERS = "-"


def partition(input_list, num_partitions):
    """Contiguous split into <= num_partitions near-equal parts, the first `remainder` parts one longer
    (reference :13-31; an empty input divides by zero there as well)."""
    if len(input_list) < num_partitions:
        num_partitions = len(input_list)
    num_per_partition = len(input_list) // num_partitions
    remainder = len(input_list) % num_partitions
    out, pos = [], 0
    for i in range(num_partitions):
        n = num_per_partition + (1 if i < remainder else 0)
        out.append(list(input_list[pos:pos + n]))
        pos += n
    return out


class ESM_MSA_sampler():
    """adapted from bert-gen bert-babble.ipynb (via pgen.esm_msa_sampler.ESM_MSA_sampler)"""

    def __init__(self, model, device="cpu"):
        self.model = model
        self.model.model = self.model.model.eval()
        self.device, self.cuda = _gibbs.resolve_device(device)
        self.model.model.to(self.device)
        self.valid_aa_idx = sorted([self.model.alphabet.get_idx(tok) for tok in ESM_MSA_ALLOWED_AMINO_ACIDS])
        self.toks = [self.model.alphabet.get_tok(idx) for idx in self.valid_aa_idx]
        self.draw_seed = None
        self.rng_stream = 0
        self.record = False
        self.last_run = []
        self.shard_over_ranks = True     # several torch.distributed ranks: generate() splits the MSAs of a batch over them

    def untokenize_batch(self, batch):
        if hasattr(batch, "tolist"):
            batch = batch.tolist()
        out_batch = list()
        for msa in batch:
            out_batch += ["".join([self.model.alphabet.get_tok(itm) for itm in seq[1:]]) for seq in msa]
        return out_batch

    def get_init_msa(self, seed_msa, max_len, batch_size=1):
        """[B, R, C] tokens: every row <mask>-padded to max_len, the MSA repeated batch_size times (reference :78-90)."""
        rows = [(str(i), _gibbs.mask_padded(seq, max_len, ESM_MSA_ALLOWED_AMINO_ACIDS)) for i, seq in enumerate(seed_msa)]
        return self.model.batch_converter([rows] * batch_size)[2]

    def clean_seed_seq(self, seq):
        return _gibbs.clean_seed(seq, ESM_MSA_ALLOWED_AMINO_ACIDS)

    def _require_gpu(self, what):
        if not self.cuda:
            raise RuntimeError("ESM_MSA_sampler.%s needs device 'gpu'/'cuda:N' on an MI355X: the Gibbs hot path is "
                               "implemented as HIP kernels only, there is no CPU implementation" % what)

    def _draw_seed(self):
        return self.draw_seed if self.draw_seed is not None else int(torch.randint(0, 2**62, (1,)).item())

    # ---- single-row resampling (reference :101-147) -------------------------------------------
    def generate_single(self, seed_msa, steps=10, passes=3, burn_in=1, target_index=0, k=1, exclude_positions=None):
        if exclude_positions is None:
            exclude_positions = []
        exclude_positions = set(i + 1 for i in exclude_positions)   # shift for the <cls> column
        self._require_gpu("generate_single")
        sequence_length = len(seed_msa[0])
        positions = [x for x in range(1, sequence_length + 1) if x not in exclude_positions]
        batch = self.get_init_msa(seed_msa, len(seed_msa[0]), 1)
        R = batch.shape[1]
        if not -R <= target_index < R:
            raise IndexError("index %d is out of bounds for dimension 0 with size %d" % (target_index, R))

        # the step lists of every pass, decided up front (selection never depends on the logits)
        from . import pyrandom
        steps_all, flags = [], []
        for pass_num in range(passes):
            pyrandom.global_shuffle(positions)
            for step in partition(positions, steps):
                steps_all.append(list(step))
                flags.append(1 if pass_num < burn_in else 0)
        n_steps = len(steps_all)
        P = max((len(s) for s in steps_all), default=0)
        table = np.full((n_steps, 1, P), -1, dtype=np.int32)
        for i, s in enumerate(steps_all):
            table[i, 0, :len(s)] = s
        tr = target_index % R
        params = _lib.make_sample_params(True, self.model.alphabet.mask_idx, k, 0, None, self.valid_aa_idx,
                                         self._draw_seed(), rng_stream=self.rng_stream, row_id_base=tr)
        if isinstance(self.model.model, NativeMaskedLM):
            tok = np.ascontiguousarray(batch.numpy(), dtype=np.int32)
            lg, st = self.model.model.gibbs_single_run(tok, R - 1, tr, table[:, 0, :], flags, params,
                                                       want_logits=self.record, want_tokens=self.record)
            batch = torch.from_numpy(tok.astype(np.int64))
            if self.record:
                self.last_run = [dict(table=table, sampled_logits=lg, sampled_tokens=st, tokens=tok.copy())]
        else:
            batch = _gibbs.run_plugin_loop(self.model.model, batch, table, params, self.device, row_map=[tr],
                                           mask_row_map=[R - 1], sample_flags=flags)
        return self.untokenize_batch(batch)[target_index]

    # ---- whole-MSA resampling (reference :151-253) ------------------------------------------------
    def generate(self, n_samples, seed_msa, batch_size=1, in_order=False, max_len=None, leader_length=0,
                 leader_length_percent=None, top_k=0, temperature=None, num_iters=10, burnin=float('inf'),
                 mask=True, num_positions=0, num_positions_percent=None, indexes=None, rollover_from_start=False,
                 show_progress_bar=True):
        num_sequences = len(seed_msa)
        sequence_length = len(seed_msa[0])
        sequences = []
        n_generation_rounds = math.ceil(n_samples / num_sequences / batch_size)
        num_positions, leader_length = _gibbs.derive_counts(sequence_length, num_positions, num_positions_percent,
                                                            leader_length, leader_length_percent)
        if max_len is None:
            max_len = sequence_length
        self._require_gpu("generate")
        draw_seed = self._draw_seed()
        native = isinstance(self.model.model, NativeMaskedLM)
        self.last_run = []
        ctx = sharding.dist_context() if (native and self.shard_over_ranks) else None
        if ctx is not None:
            sharding.sync_host_rng(ctx)
            draw_seed = sharding.broadcast_object(ctx, draw_seed)

        for generation_round in trange(n_generation_rounds, disable=(not show_progress_bar)):
            batch = self.get_init_msa(seed_msa, max_len, batch_size)        # [B, R, C]
            indexes, last_i = self.calculate_indexes(indexes, leader_length, max_len, rollover_from_start)
            indexes = _gibbs.normalise_indexes(indexes, batch.shape[2])
            if num_positions > len(indexes):
                num_positions = len(indexes)
            table, last_i = _gibbs.build_target_table(num_iters, (batch_size, num_sequences), indexes, num_positions,
                                                      in_order, last_i)
            params = _lib.make_sample_params(mask, self.model.alphabet.mask_idx, top_k, burnin, temperature,
                                             self.valid_aa_idx, draw_seed, rng_stream=self.rng_stream,
                                             row_id_base=generation_round * batch_size * num_sequences)
            if native and ctx is not None:
                def run_block(ltok, ltable, base):
                    params.row_id_base = base & 0xFFFFFFFF
                    self.model.model.set_job_items(batch.shape[0])      # shard of a batch.shape[0]-item job
                    try:
                        self.model.model.gibbs_run(ltok, ltable, params)
                    finally:
                        self.model.model.set_job_items(0)
                tok = sharding.run_sharded(ctx, np.ascontiguousarray(batch.numpy(), dtype=np.int32), table,
                                           generation_round * batch_size * num_sequences, num_sequences, run_block, self.device)
                batch = torch.from_numpy(tok.astype(np.int64))
                if self.record:
                    self.last_run.append(dict(table=table, tokens=tok.copy()))
            elif native:
                tok = np.ascontiguousarray(batch.numpy(), dtype=np.int32)
                lg, st = self.model.model.gibbs_run(tok, table, params, want_logits=self.record, want_tokens=self.record)
                batch = torch.from_numpy(tok.astype(np.int64))
                if self.record:
                    self.last_run.append(dict(table=table, sampled_logits=lg, sampled_tokens=st, tokens=tok.copy()))
            else:
                flat = table.reshape(num_iters, batch_size * num_sequences, table.shape[-1])
                batch = _gibbs.run_plugin_loop(self.model.model, batch, flat, params, self.device)
            strs = self.untokenize_batch(batch)
            if generation_round == (n_generation_rounds - 1):
                sequences += strs[0:n_samples - len(sequences)]
            else:
                sequences += strs
        return sequences

    # ---- index helpers with the reference's names (:255-304) ---------------------------------------
    def mask_target_indexes(self, batch, target_indexes):
        for batch_index in range(len(batch)):
            for sequence_index in range(len(batch[batch_index])):
                for kk in target_indexes[batch_index][sequence_index]:
                    batch[batch_index][sequence_index][kk] = self.model.alphabet.mask_idx

    def mask_target_indexes_single(self, batch, target_indexes, seq_index):
        for batch_index in range(len(batch)):
            for kk in target_indexes:
                batch[batch_index][seq_index][kk] = self.model.alphabet.mask_idx

    def get_target_indexes_all_positions(self, batch_size, indexes, num_sequences):
        return [[indexes] * num_sequences for _ in range(batch_size)]

    def get_random_target_index(self, batch_size, indexes, num_positions, num_sequences):
        from . import pyrandom
        t = pyrandom.global_sample_table(list(indexes), num_positions, batch_size * num_sequences)
        return t.reshape(batch_size, num_sequences, num_positions).tolist()

    def get_target_index_in_order(self, batch_size, indexes, next_i, num_positions, num_sequences):
        last_i, per_seq = _gibbs.in_order_window(indexes, next_i, num_positions)
        return last_i, [[per_seq] * num_sequences for _ in range(batch_size)]

    def calculate_indexes(self, indexes, leader_length, max_len, rollover_from_start):
        indexes, last_i = _gibbs.candidate_indexes(indexes, leader_length, max_len, rollover_from_start)
        return (list(indexes) if isinstance(indexes, range) else indexes), last_i     # the MSA sampler hands out a list (:296)

    # ---- masked log-likelihood of one MSA row (reference :306-432) --------------------------------------
    def log_likelihood(self, msa, target_index=0, with_masking=True, verbose=False, count_gaps=False,
                       mask_distance=float("inf")):
        return next(self.log_likelihood_batch([msa], target_index, with_masking, verbose, count_gaps, mask_distance))

    def log_likelihood_batch(self, msa_list, target_index=0, with_masking=True, verbose=False, count_gaps=False,
                             mask_distance=float("inf"), batch_size=1):
        """Same contract as pgen.esm_msa_sampler.ESM_MSA_sampler.log_likelihood_batch: yields (float mean, list[float])
        for the row `target_index` of every MSA; gap positions of the target row are skipped unless count_gaps.
        Every MSA is scored on its own tokens (no padding reaches the model)."""
        self._require_gpu("log_likelihood_batch")
        gap_tokens = {self.model.alphabet.get_idx(x) for x in ESM_MSA_GAP_CHARACTERS}
        if batch_size is None:
            batch_size = len(msa_list)
        range_start = 1 if self.model.alphabet.prepend_bos else 0
        mask_idx = self.model.alphabet.mask_idx
        for msa in msa_list:
            reformatted = [(str(i), self.clean_seed_seq(seq)) for i, seq in enumerate(msa)]
            _, _, one = self.model.batch_converter(reformatted)            # [1, R, C]
            R = one.shape[1]
            tr = target_index % R
            seq_len = len(msa[target_index])
            denom = seq_len - (0 if count_gaps else sum(msa[target_index].count(g) for g in ESM_MSA_GAP_CHARACTERS))
            end = seq_len + range_start
            orig = one[0, tr].numpy()
            if with_masking:
                n = int(min(mask_distance, seq_len))
                copies = one.repeat(n, 1, 1)
                pos_all = [list(range(range_start + i, end, n)) for i in range(n)]
                for i, pos in enumerate(pos_all):
                    copies[i, tr, pos] = mask_idx
            else:
                n = 1
                copies = one
                pos_all = [list(range(range_start, end))]
            pos_of = [[p for p in pos if count_gaps or int(orig[p]) not in gap_tokens] for pos in pos_all]
            P = max((len(p) for p in pos_of), default=0)
            idx = np.full((n, max(P, 1)), -1, dtype=np.int32)
            tgt = np.zeros((n, max(P, 1)), dtype=np.int32)
            for i, pos in enumerate(pos_of):
                idx[i, :len(pos)] = pos
                tgt[i, :len(pos)] = orig[pos]
            likelihood_sum = np.float32(0.0)
            likelihood_list = []
            for batch_start in range(0, n, max(1, batch_size)):
                sl = slice(batch_start, batch_start + max(1, batch_size))
                nb = copies[sl].shape[0]
                row_of = np.arange(nb) * R + tr
                lp = _gibbs.score_positions(self.model.model, copies[sl], row_of, idx[sl], tgt[sl], self.device)
                for i in range(nb):
                    for p in range(len(pos_of[batch_start + i])):
                        likelihood_sum = np.float32(likelihood_sum + lp[i, p])
                        likelihood_list.append(float(lp[i, p]))
            assert len(likelihood_list) == denom
            yield (float(likelihood_sum / np.float32(denom)), likelihood_list)
