"""Model configurations, synthetic weights and checkpoint reading.

The reference obtains weights by `esm.pretrained.<name>()` (/root/reference/src/pgen/models.py:61-86),
which downloads fair-esm checkpoints.  There is no network here, so:
  * `synthetic_state_dict` builds a seeded random state dict with fair-esm's key names and shapes
     -- throughput does not depend on weight values;
  * `load_fair_esm_checkpoint` reads a real fair-esm `.pt` file when one is supplied, applying the
    same prefix stripping fair-esm applies, so real ESM-1b / MSA-1b weights can drive the engine.
"""
import os
import re

import numpy as np

from . import _lib

ESM1B_CONFIG = dict(arch=_lib.PG_ARCH_ESM1B, vocab=33, d_model=1280, n_layers=33, n_heads=20, d_ffn=5120, max_positions=1024,
                    pad_idx=1, mask_idx=32, cls_idx=0, eos_idx=2, token_dropout=1, max_msa_rows=0, layer_norm_eps=1e-5)
MSA1B_CONFIG = dict(arch=_lib.PG_ARCH_MSA1B, vocab=33, d_model=768, n_layers=12, n_heads=12, d_ffn=3072, max_positions=1024,
                    pad_idx=1, mask_idx=32, cls_idx=0, eos_idx=2, token_dropout=0, max_msa_rows=1024, layer_norm_eps=1e-5)


def make_config(base, **overrides):
    cfg = dict(base)
    cfg.update(overrides)
    if "n_heads" not in overrides and "d_model" in overrides:
        cfg["n_heads"] = cfg["d_model"] // 64
    return cfg


def tensor_shapes(cfg):
    """name -> shape for every tensor the engine reads (fair-esm state-dict keys)."""
    d, f, V = cfg["d_model"], cfg["d_ffn"], cfg["vocab"]
    s = {"embed_tokens.weight": (V, d),
         "embed_positions.weight": (cfg["max_positions"] + cfg["pad_idx"] + 1, d),
         "emb_layer_norm_before.weight": (d,), "emb_layer_norm_before.bias": (d,),
         "emb_layer_norm_after.weight": (d,), "emb_layer_norm_after.bias": (d,),
         "lm_head.dense.weight": (d, d), "lm_head.dense.bias": (d,),
         "lm_head.layer_norm.weight": (d,), "lm_head.layer_norm.bias": (d,), "lm_head.bias": (V,)}

    def lin(p, o, i):
        s[p + ".weight"] = (o, i)
        s[p + ".bias"] = (o,)

    def ln(p):
        s[p + ".weight"] = (d,)
        s[p + ".bias"] = (d,)

    for i in range(cfg["n_layers"]):
        p = "layers.%d." % i
        if cfg["arch"] == _lib.PG_ARCH_ESM1B:
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                lin(p + "self_attn." + n, d, d)
            ln(p + "self_attn_layer_norm")
            lin(p + "fc1", f, d)
            lin(p + "fc2", d, f)
            ln(p + "final_layer_norm")
        else:
            for blk in ("row_self_attention", "column_self_attention"):
                for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                    lin(p + blk + ".layer." + n, d, d)
                ln(p + blk + ".layer_norm")
            lin(p + "feed_forward_layer.layer.fc1", f, d)
            lin(p + "feed_forward_layer.layer.fc2", d, f)
            ln(p + "feed_forward_layer.layer_norm")
    if cfg["arch"] == _lib.PG_ARCH_MSA1B:
        s["msa_position_embedding"] = (1, cfg["max_msa_rows"], 1, d)
    return s


def synthetic_state_dict(cfg, seed=0, std=0.02, embed_std=None, ln_jitter=0.0):
    """Seeded random weights: N(0, std^2) matrices/biases, LayerNorm gamma = 1 (+jitter), beta = 0 (+jitter)."""
    rng = np.random.default_rng(seed)
    es = std if embed_std is None else embed_std
    out = {}
    for name, shape in tensor_shapes(cfg).items():
        is_ln = "layer_norm" in name
        if is_ln and name.endswith(".weight"):
            a = 1.0 + ln_jitter * rng.standard_normal(shape, dtype=np.float32)
        elif is_ln:
            a = ln_jitter * rng.standard_normal(shape, dtype=np.float32)
        elif name.startswith("embed_") or name == "msa_position_embedding":
            a = es * rng.standard_normal(shape, dtype=np.float32)
        else:
            a = std * rng.standard_normal(shape, dtype=np.float32)
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    return out


_PREFIXES = ("encoder.sentence_encoder.", "sentence_encoder.", "encoder.", "model.")


def normalise_state_dict(sd, cfg):
    """fair-esm checkpoint keys -> the engine's keys (strip fair-esm's wrapper prefixes, untie lm_head)."""
    want = tensor_shapes(cfg)
    out = {}
    for k, v in sd.items():
        name = k
        changed = True
        while changed:
            changed = False
            for p in _PREFIXES:
                if name.startswith(p):
                    name = name[len(p):]
                    changed = True
        name = re.sub(r"^lm_head\.weight$", "embed_tokens.weight", name) if "embed_tokens.weight" not in sd else name
        if name in want:
            arr = v.detach().cpu().float().numpy() if hasattr(v, "detach") else np.asarray(v, dtype=np.float32)
            out[name] = np.ascontiguousarray(arr.reshape(want[name]), dtype=np.float32)
    missing = [k for k in want if k not in out]
    if missing:
        raise KeyError("checkpoint is missing %d tensors, e.g. %s" % (len(missing), missing[:3]))
    return out


def load_fair_esm_checkpoint(path, cfg):
    import torch
    blob = torch.load(path, map_location="cpu", weights_only=False)
    sd = blob["model"] if isinstance(blob, dict) and "model" in blob else blob
    return normalise_state_dict(sd, cfg)


def find_cached_checkpoint(filename):
    p = os.path.join(os.path.expanduser("~/.cache/torch/hub/checkpoints"), filename)
    return p if os.path.exists(p) else None
