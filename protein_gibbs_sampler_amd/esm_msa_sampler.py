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
        self.shard_over_ranks = False    # opt-in (or PGIBBS_SHARD_OVER_RANKS=1): generate() / generate_single_batch() split ONE job's MSAs over the torch.distributed ranks

    def untokenize_batch(self, batch):
        if hasattr(batch, "numpy") and getattr(batch, "ndim", 0) == 3:       # a [B, R, C] token tensor: one table lookup per row
            arr = batch.numpy()
            return self.model.alphabet.decode_rows(arr[:, :, 1:].reshape(-1, arr.shape[2] - 1))
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
        return self.generate_single_batch([seed_msa], steps=steps, passes=passes, burn_in=burn_in, target_index=target_index, k=k,
                                          exclude_positions=[exclude_positions], _shard=False)[0]

    def generate_single_batch(self, seed_msas, steps=10, passes=3, burn_in=1, target_index=0, k=1, exclude_positions=None,
                              max_batch=4, _shard=None):
        """== [self.generate_single(m, steps, passes, burn_in, target_index, k, ex) for m, ex in zip(seed_msas, exclude_positions)]
        -- the loop of pgen_msa_revised over templates and sequences per template (reference pgen_msa_revised.py:107-115) --
        with MSAs of equal shape resampled together, up to `max_batch` per native call (BASELINE config 5: batches of templates),
        and, when sharding over torch.distributed ranks is switched on, contiguous blocks of the list on different GPUs.

        Same interpreter-RNG consumption as the serial calls (one random.shuffle per MSA and pass, MSA-major), one torch seed per
        MSA in list order, and the same strings: each MSA's arithmetic is independent of its batch mates (pgibbs.h
        pg_msa_gibbs_single_batch_run).  exclude_positions: None or one list (or None) per MSA."""
        n = len(seed_msas)
        if exclude_positions is None:
            exclude_positions = [None] * n
        if len(exclude_positions) != n:
            raise ValueError("exclude_positions: expected one list (or None) per MSA")
        self._require_gpu("generate_single")
        from . import pyrandom
        native = isinstance(self.model.model, NativeMaskedLM)
        shard = sharding.sharding_requested(self.shard_over_ranks) if _shard is None else _shard
        ctx = sharding.dist_context() if (native and shard) else None
        if ctx is not None:
            if self.record:
                raise ValueError("record=True is not supported together with shard_over_ranks (per-draw logits stay on their rank)")
            sharding.check_same_job(ctx, sharding.job_digest(seed_msas, steps, passes, burn_in, target_index, k, exclude_positions,
                                                             self.rng_stream), "ESM_MSA_sampler.generate_single_batch")
            sharding.sync_host_rng(ctx)

        # every MSA's step lists, decided up front in the serial order (selection never depends on the logits)
        jobs = []
        for seed_msa, excl in zip(seed_msas, exclude_positions):
            excl = set(i + 1 for i in (excl or []))                    # shift for the <cls> column
            sequence_length = len(seed_msa[0])
            positions = [x for x in range(1, sequence_length + 1) if x not in excl]
            batch = self.get_init_msa(seed_msa, len(seed_msa[0]), 1)
            R = batch.shape[1]
            if not -R <= target_index < R:
                raise IndexError("index %d is out of bounds for dimension 0 with size %d" % (target_index, R))
            steps_all, flags = [], []
            for pass_num in range(passes):
                pyrandom.global_shuffle(positions)
                for step in partition(positions, steps):
                    steps_all.append(list(step))
                    flags.append(1 if pass_num < burn_in else 0)
            jobs.append(dict(batch=batch, steps=steps_all, flags=flags, tr=target_index % R, seed=self._draw_seed()))
        if ctx is not None:
            seeds = sharding.broadcast_object(ctx, [j["seed"] for j in jobs])      # rank 0's torch draws
            for j, sd in zip(jobs, seeds):
                j["seed"] = sd
        self.last_run = [None] * n
        mask_idx = self.model.alphabet.mask_idx

        def table_of(job, P):
            t = np.full((len(job["steps"]), P), -1, dtype=np.int32)
            for i, st in enumerate(job["steps"]):
                t[i, :len(st)] = st
            return t

        lo, hi = (0, n) if ctx is None else sharding.shard_range(n, ctx.world, ctx.rank)
        if native:
            # equal-shape MSAs of this rank's block go through the engine together
            groups = {}
            for j in range(lo, hi):
                job = jobs[j]
                key = (tuple(job["batch"].shape), len(job["steps"]), tuple(job["flags"]), job["tr"])
                groups.setdefault(key, []).append(j)
            for key, members in groups.items():
                for c0 in range(0, len(members), max(1, max_batch)):
                    chunk = members[c0:c0 + max(1, max_batch)]
                    P = max((len(st) for j in chunk for st in jobs[j]["steps"]), default=0)
                    table = np.stack([table_of(jobs[j], P) for j in chunk], axis=1)         # [n_steps, b, P]
                    tok = np.ascontiguousarray(np.concatenate([jobs[j]["batch"].numpy() for j in chunk]), dtype=np.int32)
                    R = tok.shape[1]
                    params = [_lib.make_sample_params(True, mask_idx, k, 0, None, self.valid_aa_idx, jobs[j]["seed"],
                                                      rng_stream=self.rng_stream, row_id_base=jobs[j]["tr"]) for j in chunk]
                    lg, st = self.model.model.gibbs_single_batch_run(tok, R - 1, key[3], table, list(key[2]), params,
                                                                     want_logits=self.record, want_tokens=self.record)
                    for i, j in enumerate(chunk):
                        jobs[j]["batch"] = torch.from_numpy(tok[i:i + 1].astype(np.int64))
                        if self.record:
                            self.last_run[j] = dict(table=table[:, i:i + 1], sampled_logits=lg[:, i], sampled_tokens=st[:, i],
                                                    tokens=tok[i:i + 1].copy())
        else:
            for job in jobs:
                R = job["batch"].shape[1]
                P = max((len(st) for st in job["steps"]), default=0)
                params = _lib.make_sample_params(True, mask_idx, k, 0, None, self.valid_aa_idx, job["seed"],
                                                 rng_stream=self.rng_stream, row_id_base=job["tr"])
                job["batch"] = _gibbs.run_plugin_loop(self.model.model, job["batch"], table_of(job, P)[:, None, :], params, self.device,
                                                      row_map=[job["tr"]], mask_row_map=[R - 1], sample_flags=job["flags"])
        if ctx is not None:
            # the one collective: the resampled rows, padded to the widest alignment, blocks in rank order
            width = max(j["batch"].shape[2] for j in jobs)
            mine = np.full((hi - lo, width), -1, dtype=np.int32)
            for i, j in enumerate(range(lo, hi)):
                row = jobs[j]["batch"][0, jobs[j]["tr"]].numpy()
                mine[i, :len(row)] = row
            counts = [sharding.shard_range(n, ctx.world, r)[1] - sharding.shard_range(n, ctx.world, r)[0] for r in range(ctx.world)]
            t = torch.from_numpy(mine)
            on_gpu = ctx.dist.get_backend() != "gloo"
            full = sharding.gather_tokens(ctx.dist, t.to(self.device) if on_gpu else t, counts).cpu().numpy()
            self.last_run = []                    # per-draw records are refused together with sharding (above)
            return ["".join(self.model.alphabet.get_tok(int(v)) for v in full[j, 1:jobs[j]["batch"].shape[2]]) for j in range(n)]
        if not self.record:
            self.last_run = []
        # == untokenize_batch(batch)[target_index] (reference :147) without spelling out the other rows of the alignment
        return [self.untokenize_batch(job["batch"][:, job["tr"]:job["tr"] + 1])[0] for job in jobs]

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
        ctx = sharding.dist_context() if (native and sharding.sharding_requested(self.shard_over_ranks)) else None
        if ctx is not None:
            if self.record:
                raise ValueError("record=True is not supported together with shard_over_ranks (per-draw logits stay on their rank)")
            sharding.check_same_job(ctx, sharding.job_digest(
                n_samples, seed_msa, batch_size, in_order, max_len, leader_length, top_k, temperature, num_iters, burnin, mask,
                num_positions, None if indexes is None else list(indexes), rollover_from_start, self.rng_stream), "ESM_MSA_sampler.generate")
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
                                           generation_round * batch_size * num_sequences, num_sequences, run_block, self.device, guard=self.model.model)
                batch = torch.from_numpy(tok.astype(np.int64))
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
        with_masking: every MSA is scored on its own strided-mask copies (no padding reaches the model, as in the reference,
        :365-405).  Unmasked: the reference converts the WHOLE list into one tensor padded to the deepest / widest MSA and scores
        `batch_size` of them per forward (:341, :416-431) -- so a shallower MSA of a ragged list is scored with <pad> rows and
        columns around it under fair-esm's padding semantics (and its tied row attention's 1/sqrt(R) from the padded depth); the
        same here, through the engine's <pad> handling."""
        self._require_gpu("log_likelihood_batch")
        gap_tokens = {self.model.alphabet.get_idx(x) for x in ESM_MSA_GAP_CHARACTERS}
        if batch_size is None:
            batch_size = len(msa_list)
        range_start = 1 if self.model.alphabet.prepend_bos else 0
        mask_idx = self.model.alphabet.mask_idx
        if not with_masking:
            if not msa_list:
                return
            reformatted = [[(str(i), self.clean_seed_seq(seq)) for i, seq in enumerate(msa)] for msa in msa_list]
            _, _, tokens = self.model.batch_converter(reformatted)          # [n, R_max, C_max], <pad> around the smaller MSAs
            n, R, C = tokens.shape
            # the reference's own shape check (:353): the widest target row fills the padded width
            assert max(len(msa[target_index]) for msa in msa_list) == C - range_start
            # Row scored per MSA.  target_index >= 0: row target_index, as the reference.  target_index < 0 counts from the END OF
            # EACH MSA (as the masked path below and generate_single do).  On a list of equal depths that is the reference's
            # tokens[:, target_index]; on a ragged list the reference reads that row of the PADDED tensor -- a <pad> row for every
            # shallower MSA, whose "likelihoods" are those of <pad> tokens -- a deliberate deviation.
            rows_of = [target_index if target_index >= 0 else len(msa) + target_index for msa in msa_list]
            for batch_start in range(0, n, max(1, batch_size)):
                chunk = tokens[batch_start:batch_start + max(1, batch_size)]
                nb = chunk.shape[0]
                pos_of, orig = [], []
                for i in range(nb):
                    msa = msa_list[batch_start + i]
                    o = chunk[i, rows_of[batch_start + i]].numpy()
                    end = len(msa[target_index]) + range_start
                    pos_of.append([p_ for p_ in range(range_start, end) if count_gaps or int(o[p_]) not in gap_tokens])
                    orig.append(o)
                P = max((len(p_) for p_ in pos_of), default=0)
                idx = np.full((nb, max(P, 1)), -1, dtype=np.int32)
                tgt = np.zeros((nb, max(P, 1)), dtype=np.int32)
                for i, pos in enumerate(pos_of):
                    idx[i, :len(pos)] = pos
                    tgt[i, :len(pos)] = orig[i][pos]
                row_of = np.arange(nb) * R + np.asarray(rows_of[batch_start:batch_start + nb])
                lp = _gibbs.score_positions(self.model.model, chunk, row_of, idx, tgt, self.device)
                for i in range(nb):
                    msa = msa_list[batch_start + i]
                    denom = len(msa[target_index]) - (0 if count_gaps else sum(msa[target_index].count(g) for g in ESM_MSA_GAP_CHARACTERS))
                    likelihood_sum = np.float32(0.0)
                    likelihood_list = []
                    for p_ in range(len(pos_of[i])):
                        likelihood_sum = np.float32(likelihood_sum + lp[i, p_])
                        likelihood_list.append(float(lp[i, p_]))
                    if denom == 0:                       # all-gap target row with count_gaps=False: the reference's 0.0 / 0
                        raise ZeroDivisionError("float division by zero")
                    yield (float(likelihood_sum / np.float32(denom)), likelihood_list)
            return
        for msa in msa_list:
            reformatted = [(str(i), self.clean_seed_seq(seq)) for i, seq in enumerate(msa)]
            _, _, one = self.model.batch_converter(reformatted)            # [1, R, C]
            R = one.shape[1]
            tr = target_index % R
            seq_len = len(msa[target_index])
            assert seq_len == one.shape[2] - range_start              # the reference's shape check (:353-355)
            denom = seq_len - (0 if count_gaps else sum(msa[target_index].count(g) for g in ESM_MSA_GAP_CHARACTERS))
            end = seq_len + range_start
            orig = one[0, tr].numpy()
            n = int(min(mask_distance, seq_len))
            copies = one.repeat(n, 1, 1)
            pos_all = [list(range(range_start + i, end, n)) for i in range(n)]
            for i, pos in enumerate(pos_all):
                copies[i, tr, pos] = mask_idx
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
            if denom == 0:
                raise ZeroDivisionError("float division by zero")
            yield (float(likelihood_sum / np.float32(denom)), likelihood_list)
