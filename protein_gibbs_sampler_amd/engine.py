"""`NativeMaskedLM`: the object that sits where the reference expects `model.model`.

In the reference `model.model` is a fair-esm nn.Module (`/root/reference/src/pgen/models.py:61,86`)
called as `self.model.model(batch)["logits"]` (esm_sampler.py:223, esm_msa_sampler.py:136,236) after
`.eval()` (esm_sampler.py:62) and `.to(device)` (:80).  Here it is a handle to the HIP engine
(C ABI `pg_engine_*`); besides the reference's call protocol it exposes `gibbs_run*`, which runs the
whole iteration loop on the device in one native call.
"""
import ctypes
import re

import numpy as np

from . import _lib


FP16_MAX = 65504.0


def fp16_weights_ok(state_dict):
    """Can every tensor be stored as IEEE fp16 without overflow?  (finite, |w| <= 65504; one pass, no copies)"""
    for name, a in state_dict.items():
        a = np.asarray(a)
        if a.size and not (np.isfinite(a.max()) and np.isfinite(a.min()) and max(float(a.max()), -float(a.min())) <= FP16_MAX):
            return False, name
    return True, None


class NativeMaskedLM:
    """precision: "bf16" (the benchmarked throughput mode), "fp16" (the same kernels with fp16 operands: 8x closer to the fp32
    reference for ~3 % of the speed, but no range beyond +-65504), "fp32" (strict parity mode), or "auto" (default of models.* and
    the command-line front ends since round 5) = fp16 WITH A GUARD:
      1. the weights are scanned when the object is built -- a tensor that does not fit fp16 selects bf16;
      2. when the engine is created on a GPU one small probe forward must come back finite;
      3. the shapes the fp16 MSA kernels lack (<pad> batches, alignments wider than 576 columns) are recognised from the call's
         OWN tokens before anything runs and that call goes to a second, bf16-operand engine of the same weights (built on first
         need); every later call reports non-finite logits as PG_ERR_RANGE before it overwrites anything the caller owns, and the
         host-buffer methods below then run THAT CALL again on the bf16 engine from the caller's intact inputs (one warning).
    Since round 6 the object's mode never changes after `.to()`: which operands a call runs with depends on that call
    alone -- not on what was called before it, nor on which rank of a sharded job it runs on.  Sharded jobs add one agreement step
    (sharding.run_sharded: if any rank's block overflowed, every rank runs its block in bf16, as the single-GPU call would), and a
    batched generate_single call that overflows is repeated template by template (a template's result is that of a call on it
    alone, so grouping does not matter).  The engine never produces NaN-derived draws.  Device-pointer callers (bench.py, plug-in
    loops) get the error from pg_engine_synchronize and choose themselves."""

    def __init__(self, cfg, state_dict, precision="bf16"):
        self.cfg = dict(cfg)
        self.state_dict = state_dict       # name -> float32 ndarray (host master copy)
        self.auto = precision == "auto"
        if self.auto:
            ok, worst = fp16_weights_ok(state_dict)
            precision = "fp16" if ok else "bf16"
            if not ok:
                import warnings
                warnings.warn("precision='auto': tensor %r does not fit IEEE fp16 (|w| > 65504 or not finite) -- using bf16 operands" % worst)
        self.precision = {"bf16": _lib.PG_PREC_BF16, "fp32": _lib.PG_PREC_FP32, "fp16": _lib.PG_PREC_F16}[precision]
        self._h = None
        self._alt = None                   # auto mode: the bf16-operand engine of the same weights (created on first need)
        self._job_items = 0
        self._force_bf16 = 0               # > 0: inside forced_bf16()
        self._fell_back = False            # a call since the last take_fell_back() was run again in bf16 (PG_ERR_RANGE)
        self._warned = False
        self.fallbacks = 0                 # how many calls that were
        self.device = "cpu"

    @property
    def precision_name(self):
        return {_lib.PG_PREC_BF16: "bf16", _lib.PG_PREC_FP32: "fp32", _lib.PG_PREC_F16: "fp16"}[self.precision]

    # ---- the fp16 guard of precision="auto" -------------------------------------------------------
    @property
    def auto_fp16(self):
        return bool(getattr(self, "auto", False)) and self.precision == _lib.PG_PREC_F16

    def _create(self, precision):
        m = re.match(r"^cuda:([0-9]+)$", self.device)
        L = _lib.lib()
        c = _lib.ModelConfig(**{k: self.cfg[k] for k, _ in _lib.ModelConfig._fields_})
        names = sorted(self.state_dict)
        arr = (_lib.Tensor * len(names))()
        keep = []
        for i, n in enumerate(names):
            a = np.ascontiguousarray(self.state_dict[n], dtype=np.float32)
            keep.append(a)
            arr[i].name = n.encode()
            arr[i].data = a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
            arr[i].numel = a.size
        h = ctypes.c_void_p()
        _lib.check(L.pg_engine_create(ctypes.byref(c), arr, len(names), int(m.group(1)), precision, ctypes.byref(h)))
        return h

    def _alt_handle(self):
        """The bf16-operand engine of the same weights on the same device (auto mode only), built on first need."""
        if self._alt is None:
            self.handle                              # raises the usual message when nothing is resident
            self._alt = self._create(_lib.PG_PREC_BF16)
            if self._job_items:
                _lib.check(_lib.lib().pg_engine_set_job_items(self._alt, self._job_items))
        return self._alt

    def _shape_needs_bf16(self, tok):
        """Shapes the fp16 MSA kernels lack (engine.hip: they answer PG_ERR_UNSUPPORTED): decided from the call's own tokens."""
        return bool(self.is_msa and tok is not None and (tok.shape[-1] > 576 or (tok == self.cfg["pad_idx"]).any()))

    def forced_bf16(self):
        """Context manager: calls inside run on the bf16 engine (a sharded job in which some rank's block overflowed)."""
        lm = self

        class _Forced:
            def __enter__(self_):
                lm._force_bf16 += 1

            def __exit__(self_, *exc):
                lm._force_bf16 -= 1
        return _Forced()

    def take_fell_back(self):
        f, self._fell_back = self._fell_back, False
        return f

    def _guarded(self, fn, tok=None, inout=None, per_item=None):
        """Run fn(handle).  Auto mode on fp16 operands: a call whose shape the fp16 kernels lack goes to the bf16 engine straight
        away; a range error (or the fp16-specific unsupported-shape answer, should the shape test above ever miss one) runs THIS
        call again on the bf16 engine, with the in/out buffer restored to what the caller passed.  per_item (optional): how to
        repeat a batched call item by item instead (generate_single batches: only the overflowing templates move to bf16)."""
        if not self.auto_fp16:
            return fn(self.handle)
        if self._force_bf16 or self._shape_needs_bf16(tok):
            return fn(self._alt_handle())
        keep = inout.copy() if inout is not None else None
        try:
            return fn(self.handle)
        except _lib.PgError as e:
            fp16_shape = e.code == _lib.PG_ERR_UNSUPPORTED and e.msg.startswith("fp16 precision mode")
            if e.code != _lib.PG_ERR_RANGE and not fp16_shape:
                raise
            if not self._warned:
                import warnings
                self._warned = True
                warnings.warn("precision='auto': the fp16-operand engine reported %r -- this call is run again with bf16 operands "
                              "(the engine keeps fp16 operands for calls that fit; further such calls are not announced)" % (e.msg,))
            if keep is not None:
                inout[...] = keep
            if e.code == _lib.PG_ERR_RANGE:
                self._fell_back = True
                self.fallbacks += 1
            if per_item is not None:
                return per_item()
            return fn(self._alt_handle())

    def _check_fp16_msa_shape(self, tok):
        """Explicit precision="fp16" on the MSA engine: the shapes its fp16 kernels lack are refused BEFORE any work is queued
        (they used to surface as PG_ERR_UNSUPPORTED in the middle of a pgen_msa run).  precision="auto" goes on and
        falls back to bf16 inside _guarded."""
        if self.is_msa and self.precision == _lib.PG_PREC_F16 and not self.auto:
            if tok.shape[-1] > 576:
                raise ValueError("precision='fp16': alignments wider than 576 token columns (<cls> + 575 residues) take the row "
                                 "attention's scores-through-scratch form, which exists for bf16 operands only -- use "
                                 "precision='auto', 'bf16' or 'fp32'")
            if (tok == self.cfg["pad_idx"]).any():
                raise ValueError("precision='fp16': batches that hold <pad> (ragged MSA lists) are supported with bf16 operands "
                                 "only -- use precision='auto', 'bf16' or 'fp32'")

    def _probe(self):
        """One small forward right after the engine exists: a checkpoint whose activations leave the fp16 range is moved to bf16
        here, before any user call (PGIBBS_F16_PROBE=0 skips it)."""
        import os
        if os.environ.get("PGIBBS_F16_PROBE", "1") in ("", "0"):
            return
        c = self.cfg
        n = max(4, min(64, c["max_positions"] - 2))
        body = 4 + (np.arange(2 * n).reshape(2, n) * 7 + 3) % 20           # the 20 amino-acid ids 4..23, deterministic
        body[0, 1::9] = c["mask_idx"]
        if self.is_msa:
            tok = np.concatenate([np.full((2, 1), c["cls_idx"]), body], axis=1)[None].repeat(2, axis=1).reshape(1, 4, n + 1)
        elif c["arch"] == _lib.PG_ARCH_ESM1:
            tok = np.concatenate([np.full((2, 1), c["cls_idx"]), body], axis=1)
        else:
            tok = np.concatenate([np.full((2, 1), c["cls_idx"]), body, np.full((2, 1), c["eos_idx"])], axis=1)
        try:
            L = _lib.lib()
            if self.is_msa:
                B, R, C = tok.shape
                out = np.empty((B, R, C, self.cfg["vocab"]), dtype=np.float32)
                t32 = np.ascontiguousarray(tok, dtype=np.int32)
                _lib.check(L.pg_msa_forward_logits(self._h, _lib.ptr(t32), B, R, C, _lib.ptr(out)))
            else:
                B, T = tok.shape
                out = np.empty((B, T, self.cfg["vocab"]), dtype=np.float32)
                t32 = np.ascontiguousarray(tok, dtype=np.int32)
                _lib.check(L.pg_esm_forward_logits(self._h, _lib.ptr(t32), B, T, _lib.ptr(out)))
        except _lib.PgError as e:
            if e.code != _lib.PG_ERR_RANGE:
                raise
            # a property of the WEIGHTS, found before any job runs (the same on every rank of a sharded job): bf16 for good
            import warnings
            warnings.warn("precision='auto': the fp16-operand engine reported %r in the probe forward -- rebuilding it with bf16 "
                          "operands; this model runs in bf16" % (e.msg,))
            h, self._h = self._h, None
            _lib.lib().pg_engine_destroy(h)
            self.precision = _lib.PG_PREC_BF16
            self._h = self._create(self.precision)

    # ---- nn.Module-ish protocol used by the samplers -------------------------------------
    def eval(self):
        return self

    def to(self, device):
        device = str(device)
        if device == self.device and (self._h is not None or device == "cpu"):
            return self
        self._destroy()
        self.device = device
        if device == "cpu":
            return self       # host-only object: tokenisation/index helpers work, forward() raises
        m = re.match(r"^cuda:([0-9]+)$", device)
        if not m:
            raise Exception("Invalid device: " + device)
        self._h = self._create(self.precision)
        if self._job_items:
            _lib.check(_lib.lib().pg_engine_set_job_items(self._h, self._job_items))
        if self.auto and self.precision == _lib.PG_PREC_F16:
            self._probe()
        return self

    def cuda(self, device=0):
        return self.to("cuda:%d" % device if isinstance(device, int) else device)

    def _destroy(self):
        for attr in ("_h", "_alt"):
            h = getattr(self, attr, None)
            setattr(self, attr, None)
            if h:
                _lib.lib().pg_engine_destroy(h)

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    @property
    def handle(self):
        if self._h is None:
            raise RuntimeError("the HIP engine is not resident on a GPU (device=%r): call .to('cuda:N') on an MI355X; "
                               "there is no CPU implementation of the forward pass" % self.device)
        return self._h

    @property
    def is_msa(self):
        return self.cfg["arch"] == _lib.PG_ARCH_MSA1B

    # ---- forward: tokens -> {"logits": ...} ------------------------------------------------
    def __call__(self, tokens):
        import torch
        t = tokens.detach().cpu().numpy() if hasattr(tokens, "detach") else np.asarray(tokens)
        logits = self.forward_logits(t)
        out = torch.from_numpy(logits)
        if hasattr(tokens, "device") and tokens.device.type == "cuda":
            out = out.to(tokens.device)
        return {"logits": out}

    def forward_logits(self, tokens):
        tok = np.ascontiguousarray(tokens, dtype=np.int32)
        V = self.cfg["vocab"]
        L = _lib.lib()
        if self.is_msa:
            B, R, C = tok.shape
            self._check_fp16_msa_shape(tok)
            out = np.empty((B, R, C, V), dtype=np.float32)
            self._guarded(lambda h: _lib.check(L.pg_msa_forward_logits(h, _lib.ptr(tok), B, R, C, _lib.ptr(out))), tok)
        else:
            B, T = tok.shape
            out = np.empty((B, T, V), dtype=np.float32)
            self._guarded(lambda h: _lib.check(L.pg_esm_forward_logits(h, _lib.ptr(tok), B, T, _lib.ptr(out))), tok)
        return out

    def forward_logprobs(self, tokens, row_of, idx, targets):
        """log_softmax(logits)[target] at positions idx[s] of token row row_of[s] (C ABI pg_*_forward_logprobs)."""
        tok = np.ascontiguousarray(tokens, dtype=np.int32)
        row_of = np.ascontiguousarray(row_of, dtype=np.int32)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        targets = np.ascontiguousarray(targets, dtype=np.int32)
        n_sel, P = idx.shape
        out = np.zeros((n_sel, P), dtype=np.float32)
        L = _lib.lib()
        if self.is_msa:
            B, R, C = tok.shape
            self._check_fp16_msa_shape(tok)
            self._guarded(lambda h: _lib.check(L.pg_msa_forward_logprobs(h, _lib.ptr(tok), B, R, C, _lib.ptr(row_of), _lib.ptr(idx),
                                                                         _lib.ptr(targets), n_sel, P, _lib.ptr(out))), tok)
        else:
            B, T = tok.shape
            self._guarded(lambda h: _lib.check(L.pg_esm_forward_logprobs(h, _lib.ptr(tok), B, T, _lib.ptr(row_of), _lib.ptr(idx),
                                                                         _lib.ptr(targets), n_sel, P, _lib.ptr(out))), tok)
        return out

    # ---- whole Gibbs loops -----------------------------------------------------------------
    def gibbs_run(self, tokens, target_idx, params, want_logits=False, want_tokens=False):
        """tokens int32 [B,T] or [B,R,C] (modified in place); target_idx int32 [iters, B, P] / [iters, B, R, P]."""
        tok = tokens
        assert tok.dtype == np.int32 and tok.flags.c_contiguous
        idx = np.ascontiguousarray(target_idx, dtype=np.int32)
        n_iters, P = idx.shape[0], idx.shape[-1]
        V = self.cfg["vocab"]
        lg = np.empty(idx.shape + (V,), dtype=np.float32) if want_logits else None
        st = np.empty(idx.shape, dtype=np.int32) if want_tokens else None
        L = _lib.lib()
        if self.is_msa:
            B, R, C = tok.shape
            self._check_fp16_msa_shape(tok)
            self._guarded(lambda h: _lib.check(L.pg_msa_gibbs_run(h, _lib.ptr(tok), B, R, C, _lib.ptr(idx), n_iters, P, ctypes.byref(params),
                                                                  _lib.ptr(lg) if want_logits else None, _lib.ptr(st) if want_tokens else None)), tok, tok)
        else:
            B, T = tok.shape
            self._guarded(lambda h: _lib.check(L.pg_esm_gibbs_run(h, _lib.ptr(tok), B, T, _lib.ptr(idx), n_iters, P, ctypes.byref(params),
                                                                  _lib.ptr(lg) if want_logits else None, _lib.ptr(st) if want_tokens else None)), tok, tok)
        return lg, st

    def gibbs_single_run(self, tokens, mask_row, target_row, step_idx, step_sample, params, want_logits=False,
                         want_tokens=False):
        """generate_single on one MSA: tokens int32 [1,R,C] (in place), step_idx [n_steps, P]."""
        assert tokens.ndim == 3 and tokens.shape[0] == 1
        idx = np.ascontiguousarray(step_idx, dtype=np.int32)
        lg, st = self.gibbs_single_batch_run(tokens, mask_row, target_row, idx[:, None, :], step_sample, [params], want_logits,
                                             want_tokens)
        return (lg[:, 0] if lg is not None else None), (st[:, 0] if st is not None else None)

    def gibbs_single_batch_run(self, tokens, mask_row, target_row, step_idx, step_sample, params_list, want_logits=False,
                               want_tokens=False):
        """generate_single on B MSAs of equal shape in one pass (C ABI pg_msa_gibbs_single_batch_run): tokens int32 [B,R,C]
        (in place), step_idx int32 [n_steps, B, P] (template b's own partitions, -1 padded), step_sample [n_steps],
        params_list: one SampleParams per template."""
        tok = tokens
        assert tok.dtype == np.int32 and tok.flags.c_contiguous and tok.ndim == 3
        B, R, C = tok.shape
        self._check_fp16_msa_shape(tok)
        idx = np.ascontiguousarray(step_idx, dtype=np.int32)
        flags = np.ascontiguousarray(step_sample, dtype=np.int32)
        n_steps, Bi, P = idx.shape
        assert Bi == B and len(params_list) == B and flags.shape == (n_steps,)
        arr = (_lib.SampleParams * B)(*params_list)
        V = self.cfg["vocab"]
        lg = np.empty((n_steps, B, P, V), dtype=np.float32) if want_logits else None
        st = np.empty((n_steps, B, P), dtype=np.int32) if want_tokens else None

        def one_by_one():
            # a batched call overflowed: template b's result is that of a call on it alone (bit for bit), so repeating the batch
            # template by template moves exactly the overflowing templates to bf16 -- whatever the grouping or the sharding was
            for b in range(B):
                tb = np.ascontiguousarray(tok[b:b + 1])
                lb, sb = self.gibbs_single_batch_run(tb, mask_row, target_row, np.ascontiguousarray(idx[:, b:b + 1]), flags,
                                                     [params_list[b]], want_logits, want_tokens)
                tok[b:b + 1] = tb
                if want_logits:
                    lg[:, b:b + 1] = lb
                if want_tokens:
                    st[:, b:b + 1] = sb
        self._guarded(lambda h: _lib.check(_lib.lib().pg_msa_gibbs_single_batch_run(
            h, _lib.ptr(tok), B, R, C, mask_row, target_row, _lib.ptr(idx), _lib.ptr(flags), n_steps, P, arr,
            _lib.ptr(lg) if want_logits else None, _lib.ptr(st) if want_tokens else None)), tok, tok, one_by_one if B > 1 else None)
        return lg, st

    def get_stat(self, name):
        v = ctypes.c_int64(0)
        _lib.check(_lib.lib().pg_engine_get_stat(self.handle, name.encode(), ctypes.byref(v)))
        return v.value

    # ---- measurement -------------------------------------------------------------------------
    def prof_enable(self, on=True):
        _lib.check(_lib.lib().pg_prof_enable(self.handle, 1 if on else 0))

    def prof_reset(self):
        _lib.check(_lib.lib().pg_prof_reset(self.handle))

    def prof_get_kernels(self, kernel_class):
        """The kernels the GEMM dispatch picked for the profiled launches of `kernel_class` (pg_prof_get_kernels)."""
        buf = ctypes.create_string_buffer(1024)
        _lib.check(_lib.lib().pg_prof_get_kernels(self.handle, kernel_class.encode(), buf, len(buf)))
        return buf.value.decode()

    def prof_get(self, kernel_class):
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        _lib.check(_lib.lib().pg_prof_get(self.handle, kernel_class.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def synchronize(self):
        _lib.check(_lib.lib().pg_engine_synchronize(self.handle))

    def set_job_items(self, n):
        """Batch items (chains / MSAs) of the whole multi-GPU job the following calls are shards of; 0 = whole jobs again
        (pgibbs.h pg_engine_set_job_items: keeps every shard bit-identical with the single-GPU run)."""
        self._job_items = int(n)
        _lib.check(_lib.lib().pg_engine_set_job_items(self.handle, int(n)))
        if self._alt is not None:
            _lib.check(_lib.lib().pg_engine_set_job_items(self._alt, int(n)))
