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


# ESM-1 (fair-esm "protein_bert_base": esm1_t6_43M / t12_85M / t34_670M_UR50S -- pgen.models.ESM6 / ESM12 / ESM34): sqrt(d) embedding
# scale, sinusoidal positions, no emb_layer_norm_before / after, bias_k / bias_v, LayerNorm eps 1e-12, untied embed_out, no token
# dropout, the 35-token "ESM-1" alphabet (include/pgibbs.h PG_ARCH_ESM1)
ESM1_T6_CONFIG = dict(arch=_lib.PG_ARCH_ESM1, vocab=35, d_model=768, n_layers=6, n_heads=12, d_ffn=3072, max_positions=1024,
                      pad_idx=1, mask_idx=33, cls_idx=32, eos_idx=2, token_dropout=0, max_msa_rows=0, layer_norm_eps=1e-12)
ESM1_T12_CONFIG = dict(ESM1_T6_CONFIG, n_layers=12)
ESM1_T34_CONFIG = dict(ESM1_T6_CONFIG, n_layers=34, d_model=1280, n_heads=20, d_ffn=5120)


def sinusoidal_positions(n_rows, d, pad_idx):
    """fairseq / fair-esm SinusoidalPositionalEmbedding.get_embedding in float32 ([sin | cos] halves, inv_freq =
    exp(-i log(10000) / (d/2 - 1)), padding row zero): the engine reads it as its `embed_positions.weight` table."""
    half = d // 2
    step = np.float32(np.log(10000.0) / (half - 1))
    inv = np.exp(np.arange(half, dtype=np.float32) * -step).astype(np.float32)
    ang = (np.arange(n_rows, dtype=np.float32)[:, None] * inv[None, :]).astype(np.float32)
    emb = np.concatenate([np.sin(ang), np.cos(ang)], axis=1).astype(np.float32)
    if d % 2:
        emb = np.concatenate([emb, np.zeros((n_rows, 1), np.float32)], axis=1)
    emb[pad_idx] = 0
    return emb


def make_config(base, **overrides):
    cfg = dict(base)
    cfg.update(overrides)
    if "n_heads" not in overrides and "d_model" in overrides:
        cfg["n_heads"] = cfg["d_model"] // 64
    return cfg


def tensor_shapes(cfg):
    """name -> shape for every tensor the engine reads (fair-esm state-dict keys)."""
    d, f, V = cfg["d_model"], cfg["d_ffn"], cfg["vocab"]
    if cfg["arch"] == _lib.PG_ARCH_ESM1:
        s = {"embed_tokens.weight": (V, d), "embed_positions.weight": (cfg["max_positions"] + cfg["pad_idx"] + 1, d),
             "embed_out.weight": (V, d), "embed_out.bias": (V,)}
    else:
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
        if cfg["arch"] in (_lib.PG_ARCH_ESM1B, _lib.PG_ARCH_ESM1):
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                lin(p + "self_attn." + n, d, d)
            if cfg["arch"] == _lib.PG_ARCH_ESM1:
                s[p + "self_attn.bias_k"] = (d,)
                s[p + "self_attn.bias_v"] = (d,)
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
        elif name == "embed_positions.weight" and cfg["arch"] == _lib.PG_ARCH_ESM1:
            a = sinusoidal_positions(shape[0], shape[1], cfg["pad_idx"])
        elif name.startswith("embed_") or name == "msa_position_embedding":
            a = es * rng.standard_normal(shape, dtype=np.float32)
        elif name.endswith("bias_k") or name.endswith("bias_v"):
            a = 0.3 * rng.standard_normal(shape, dtype=np.float32)
        else:
            a = std * rng.standard_normal(shape, dtype=np.float32)
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    return out


def _strip_fair_esm_prefixes(name):
    """fair-esm's `_load_model_and_alphabet_core_v1` (esm/pretrained.py, [recalled]: the package is not installed here) drops
    everything up to and including "encoder." and "sentence_encoder." from v1 checkpoint keys
    (`encoder.sentence_encoder.layers.0...` -> `layers.0...`, `encoder.lm_head.dense.weight` -> `lm_head.dense.weight`)."""
    for marker in ("sentence_encoder.", "encoder."):
        if marker in name:
            name = name.split(marker, 1)[1]
    if name.startswith("decoder."):            # ESM-1 ("protein_bert_base") checkpoints: everything under `decoder.`
        name = name[len("decoder."):]
    return name[len("model."):] if name.startswith("model.") else name


def _swap_row_column(name):
    """MSA Transformer checkpoints only: fair-esm's loader exchanges "row" and "column" in every key (its lambda `prs3`):
    in esm_msa1b_t12_100M_UR50S.pt the tensors stored as `row_self_attention.*` belong to the module's
    `column_self_attention` and vice versa.  All of them are d x d, so loading them unswapped would not fail -- it would
    silently put the tied-row weights into column attention."""
    if "row" in name:
        return name.replace("row", "column")
    return name.replace("column", "row")


def normalise_state_dict(sd, cfg, fair_esm_layout=True):
    """fair-esm checkpoint keys -> the engine's keys.

    fair_esm_layout=True applies what fair-esm's own loader does to an on-disk checkpoint: prefix stripping, and for the
    MSA Transformer the row<->column swap.  Pass False for a state dict that already has module names (e.g. one taken
    from a constructed fair-esm model with `model.state_dict()`, or `synthetic_state_dict`).
    Shapes are checked exactly; a tied decoder (`lm_head.weight`) must equal `embed_tokens.weight`."""
    want = tensor_shapes(cfg)
    is_msa = cfg["arch"] == _lib.PG_ARCH_MSA1B
    named = {}
    for k, v in sd.items():
        name = k
        if fair_esm_layout:
            name = _strip_fair_esm_prefixes(name)
            if is_msa:
                name = _swap_row_column(name)
        arr = v.detach().cpu().float().numpy() if hasattr(v, "detach") else np.asarray(v, dtype=np.float32)
        if cfg["arch"] == _lib.PG_ARCH_ESM1:
            # fair-esm's ESM-1 module names -> the engine's: the untied output projection is a bare Parameter pair, bias_k / bias_v
            # are stored [1, 1, d]; the sinusoidal table is a buffer (`embed_positions._float_tensor`) that is regenerated here
            name = {"embed_out": "embed_out.weight", "embed_out_bias": "embed_out.bias"}.get(name, name)
            if name.endswith("self_attn.bias_k") or name.endswith("self_attn.bias_v"):
                arr = arr.reshape(-1)
        named[name] = arr
    if cfg["arch"] == _lib.PG_ARCH_ESM1 and "embed_positions.weight" not in named:
        named["embed_positions.weight"] = sinusoidal_positions(cfg["max_positions"] + cfg["pad_idx"] + 1, cfg["d_model"], cfg["pad_idx"])
    if cfg["arch"] == _lib.PG_ARCH_ESM1 and "embed_out.bias" not in named and "embed_out.weight" in named:
        named["embed_out.bias"] = np.zeros(cfg["vocab"], dtype=np.float32)          # final_bias = False checkpoints
    if "lm_head.weight" in named and cfg["arch"] != _lib.PG_ARCH_ESM1:
        if "embed_tokens.weight" not in named:
            named["embed_tokens.weight"] = named["lm_head.weight"]
        elif not np.array_equal(named["lm_head.weight"], named["embed_tokens.weight"]):
            raise ValueError("checkpoint has an untied lm_head.weight: this engine implements the tied decoder of ESM-1b / MSA-1b")
    out = {}
    for name, shape in want.items():
        if name not in named:
            continue
        arr = named[name]
        if tuple(arr.shape) != tuple(shape):
            raise ValueError("tensor %r has shape %s, expected %s" % (name, tuple(arr.shape), tuple(shape)))
        out[name] = np.ascontiguousarray(arr, dtype=np.float32)
    missing = [k for k in want if k not in out]
    if missing:
        raise KeyError("checkpoint is missing %d tensors, e.g. %s" % (len(missing), missing[:3]))
    if fair_esm_layout and cfg.get("token_dropout"):
        # fair-esm zeroes the <mask> embedding row when it loads an ESM-1b checkpoint ("For token drop", [recalled]); the
        # forward never reads that row as an input (masked positions are zeroed), but the tied decoder does: after loading,
        # logit[<mask>] = lm_head.bias[<mask>] only.  The samplers never draw <mask> (it is not in valid_aa_idx) and
        # log_likelihood gathers the log-softmax at real residues, whose normaliser includes that bias-only term -- exactly
        # what fair-esm computes after the same zeroing.  [recalled]: not checkable offline; a real checkpoint file decides.
        out["embed_tokens.weight"] = out["embed_tokens.weight"].copy()
        out["embed_tokens.weight"][cfg["mask_idx"]] = 0.0
    return out


def load_fair_esm_checkpoint(path, cfg):
    """Read a fair-esm `.pt` file ({"args"/"cfg": ..., "model": state dict}) as the reference does through
    `esm.pretrained.*` (/root/reference/src/pgen/models.py:61,86)."""
    import torch
    blob = torch.load(path, map_location="cpu", weights_only=False)
    sd = blob["model"] if isinstance(blob, dict) and "model" in blob else blob
    arch = None
    if isinstance(blob, dict):
        a = blob.get("args")
        arch = a.get("arch") if isinstance(a, dict) else getattr(a, "arch", None)
        if arch is None and isinstance(blob.get("cfg"), dict):
            arch = blob["cfg"].get("model", {}).get("arch") if isinstance(blob["cfg"].get("model"), dict) else None
    want_msa = cfg["arch"] == _lib.PG_ARCH_MSA1B
    if arch is not None and (arch == "msa_transformer") != want_msa:
        raise ValueError("checkpoint arch %r does not match the requested %s engine" % (arch, "MSA-1b" if want_msa else "ESM-1b"))
    if arch is not None and (arch == "protein_bert_base") != (cfg["arch"] == _lib.PG_ARCH_ESM1):
        raise ValueError("checkpoint arch %r does not match the requested %s engine"
                         % (arch, "ESM-1" if cfg["arch"] == _lib.PG_ARCH_ESM1 else "ESM-1b / MSA-1b"))
    return normalise_state_dict(sd, cfg, fair_esm_layout=True)


def to_fair_esm_checkpoint_layout(sd, cfg):
    """Inverse of the key mapping above: engine/module names -> the on-disk fair-esm v1 key layout (used to write test
    fixtures and to export weights): `encoder.sentence_encoder.` prefix on trunk tensors, `encoder.` on the LM head, and
    row<->column exchanged for the MSA Transformer."""
    is_msa = cfg["arch"] == _lib.PG_ARCH_MSA1B
    out = {}
    for name, v in sd.items():
        k = _swap_row_column(name) if is_msa else name
        k = ("encoder." + k) if k.startswith("lm_head.") else ("encoder.sentence_encoder." + k)
        out[k] = v
    return out


def find_cached_checkpoint(filename):
    p = os.path.join(os.path.expanduser("~/.cache/torch/hub/checkpoints"), filename)
    return p if os.path.exists(p) else None
