"""ctypes binding of the C ABI declared in include/pgibbs.h.

There is exactly one implementation of the hot path -- the HIP library.  If it is missing, this
module raises at import/first use; nothing here (or anywhere in the package) falls back to a CPU
or PyTorch implementation.
"""
import ctypes
import math
import os
import sys
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint32, c_uint64,
                    c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PGIBBS_LIB_PATH") or os.path.join(_HERE, "lib", "libpgibbs.so")     # override: A/B runs of two builds

PG_OK = 0
PG_ERR_INVALID, PG_ERR_HIP, PG_ERR_NO_DEVICE, PG_ERR_WEIGHTS, PG_ERR_UNSUPPORTED, PG_ERR_RANGE = 1, 2, 3, 4, 5, 6
PG_ARCH_ESM1B, PG_ARCH_MSA1B, PG_ARCH_ESM1 = 1, 2, 3
PG_COMM_ID_BYTES = 128
PG_PREC_BF16, PG_PREC_FP32, PG_PREC_F16 = 0, 1, 2
INT32_MAX = 2**31 - 1


class PgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("pgibbs error %d: %s" % (code, msg))
        self.code = code
        self.msg = msg


class ModelConfig(Structure):
    _fields_ = [("arch", c_int32), ("vocab", c_int32), ("d_model", c_int32), ("n_layers", c_int32), ("n_heads", c_int32),
                ("d_ffn", c_int32), ("max_positions", c_int32), ("pad_idx", c_int32), ("mask_idx", c_int32),
                ("cls_idx", c_int32), ("eos_idx", c_int32), ("token_dropout", c_int32), ("max_msa_rows", c_int32),
                ("layer_norm_eps", c_float)]


class Tensor(Structure):
    _fields_ = [("name", c_char_p), ("data", POINTER(c_float)), ("numel", c_int64)]


class SampleParams(Structure):
    _fields_ = [("mask", c_int32), ("mask_idx", c_int32), ("top_k", c_int32), ("burnin", c_int32),
                ("temperature", c_float), ("n_valid", c_int32), ("valid_idx", c_int32 * 32), ("rng_seed", c_uint64),
                ("rng_stream", c_uint32), ("row_id_base", c_uint32), ("iter_base", c_int32)]


_lib = None

# (name, restype, argtypes) -- one row per symbol of include/pgibbs.h
_P32 = POINTER(c_int32)
_PF = POINTER(c_float)
_PU32 = POINTER(c_uint32)
SIGNATURES = [
    ("pg_version", c_char_p, []),
    ("pg_last_error", c_char_p, []),
    ("pg_device_count", c_int, []),
    ("pg_pyrandom_create", c_void_p, []),
    ("pg_pyrandom_destroy", None, [c_void_p]),
    ("pg_pyrandom_seed", c_int, [c_void_p, _PU32, c_int]),
    ("pg_pyrandom_setstate", c_int, [c_void_p, _PU32, c_int]),
    ("pg_pyrandom_getstate", c_int, [c_void_p, _PU32, POINTER(c_int)]),
    ("pg_pyrandom_getrandbits32", c_uint32, [c_void_p, c_int]),
    ("pg_pyrandom_random", c_double, [c_void_p]),
    ("pg_pyrandom_sample", c_int, [c_void_p, _P32, c_int, c_int, _P32]),
    ("pg_pyrandom_sample_table", c_int, [c_void_p, _P32, c_int, c_int, c_int64, _P32]),
    ("pg_pyrandom_shuffle", c_int, [c_void_p, _P32, c_int]),
    ("pg_pyrandom_choices", c_int, [c_void_p, c_int, c_int, _P32]),
    ("pg_engine_create", c_int, [POINTER(ModelConfig), POINTER(Tensor), c_int, c_int, c_int, POINTER(c_void_p)]),
    ("pg_engine_destroy", None, [c_void_p]),
    ("pg_engine_set_stream", c_int, [c_void_p, c_void_p]),
    ("pg_engine_synchronize", c_int, [c_void_p]),
    ("pg_engine_device", c_int, [c_void_p]),
    ("pg_engine_set_job_items", c_int, [c_void_p, c_int64]),
    ("pg_engine_get_stat", c_int, [c_void_p, c_char_p, POINTER(c_int64)]),
    ("pg_esm_forward_logits", c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    ("pg_esm_gibbs_run", c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, POINTER(SampleParams), c_void_p,
                                 c_void_p]),
    ("pg_esm_gibbs_run_device", c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, POINTER(SampleParams),
                                        c_void_p, c_void_p]),
    ("pg_msa_forward_logits", c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    ("pg_msa_gibbs_run", c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, POINTER(SampleParams),
                                 c_void_p, c_void_p]),
    ("pg_msa_gibbs_run_device", c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, POINTER(SampleParams),
                                        c_void_p, c_void_p]),
    ("pg_msa_gibbs_single_run", c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                        POINTER(SampleParams), c_void_p, c_void_p]),
    ("pg_msa_gibbs_single_batch_run", c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                              c_int, POINTER(SampleParams), c_void_p, c_void_p]),
    ("pg_esm_forward_logprobs", c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    ("pg_msa_forward_logprobs", c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                        c_void_p]),
    ("pg_logprob_gather_device", c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                         c_void_p]),
    ("pg_mask_scatter_device", c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_int, c_int]),
    ("pg_sample_writeback_device", c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int64,
                                           c_int, POINTER(SampleParams), c_int, c_void_p]),
    ("pg_comm_unique_id", c_int, [c_void_p]),
    ("pg_comm_create", c_int, [c_int, c_int, c_void_p, c_int, POINTER(c_void_p)]),
    ("pg_comm_destroy", None, [c_void_p]),
    ("pg_comm_rank", c_int, [c_void_p]),
    ("pg_comm_world", c_int, [c_void_p]),
    ("pg_gather_tokens", c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, POINTER(c_int64), c_void_p]),
    ("pg_dbg_gather_tokens_host", c_int, [c_int, c_int, c_void_p, c_int64, c_int, POINTER(c_int64), c_int, c_void_p, c_void_p, c_void_p]),
    ("pg_dbg_gather_plan", c_int, [c_int, c_int, c_int64, c_int, POINTER(c_int64), c_int, POINTER(c_int64), POINTER(c_int64)]),
    ("pg_prof_enable", c_int, [c_void_p, c_int]),
    ("pg_prof_reset", c_int, [c_void_p]),
    ("pg_prof_get", c_int, [c_void_p, c_char_p, POINTER(c_double), POINTER(c_int64)]),
    ("pg_prof_get_kernels", c_int, [c_void_p, c_char_p, c_char_p, c_int]),
    ("pg_dbg_gemm", c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int]),
    ("pg_dbg_gemm_bench", c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_double)]),
    ("pg_dbg_rowln_bench", c_int, [c_int, c_int, c_int, c_int, POINTER(c_double), POINTER(c_double)]),
    ("pg_dbg_qkv_attention_bench", c_int, [c_int, c_int, c_int, c_int, c_int, POINTER(c_double), POINTER(c_double)]),
    ("pg_dbg_layernorm", c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float]),
    ("pg_dbg_attention", c_int, [c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int]),
    ("pg_dbg_msa_attention", c_int, [c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float]),
]


def lib():
    """Load libpgibbs.so once.  torch (when installed) is imported first so that its bundled HIP
    runtime (same SONAME, libamdhip64.so.7) is the single runtime of the process."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "protein_gibbs_sampler_amd: the HIP library %s is missing. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C protein_gibbs_sampler_amd/csrc`). "
            "There is no CPU fallback." % LIB_PATH)
    if os.environ.get("PGIBBS_HIP_RUNTIME", "auto") != "system":
        try:
            import torch  # noqa: F401  (loads torch/lib/libamdhip64.so first)
        except ImportError:
            pass
    handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, restype, argtypes in SIGNATURES:
        fn = getattr(handle, name)   # AttributeError here == header/library mismatch: fail loudly
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = handle
    return _lib


def check(rc):
    if rc != PG_OK:
        msg = lib().pg_last_error()
        raise PgError(rc, msg.decode() if msg else "")


def ptr(arr):
    """numpy array -> void* (keeps no reference: caller holds the array)."""
    return arr.ctypes.data_as(c_void_p)


def make_sample_params(mask, mask_idx, top_k, burnin, temperature, valid_idx, rng_seed, rng_stream=0, row_id_base=0,
                       iter_base=0):
    p = SampleParams()
    p.mask = 1 if mask else 0
    p.mask_idx = int(mask_idx)
    p.top_k = int(max(min(top_k, INT32_MAX), -INT32_MAX))
    # sample = (ii < burnin) with integer ii  <=>  ii < ceil(burnin)
    p.burnin = INT32_MAX if burnin == float("inf") or burnin > INT32_MAX else int(max(math.ceil(burnin), -INT32_MAX))
    p.temperature = float("nan") if temperature is None else float(temperature)
    if not 1 <= len(valid_idx) <= 32:
        raise ValueError("valid_idx must have 1..32 entries")
    p.n_valid = len(valid_idx)
    for i, v in enumerate(valid_idx):
        p.valid_idx[i] = int(v)
    p.rng_seed = int(rng_seed) & (2**64 - 1)
    p.rng_stream = int(rng_stream) & 0xFFFFFFFF
    p.row_id_base = int(row_id_base) & 0xFFFFFFFF
    p.iter_base = int(iter_base)
    return p
