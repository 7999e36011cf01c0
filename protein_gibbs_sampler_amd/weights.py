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


_ARCH_OF = {"roberta_large": _lib.PG_ARCH_ESM1B, "protein_bert_base": _lib.PG_ARCH_ESM1, "msa_transformer": _lib.PG_ARCH_MSA1B}
_ARCH_NAME = {_lib.PG_ARCH_ESM1B: "ESM-1b", _lib.PG_ARCH_ESM1: "ESM-1", _lib.PG_ARCH_MSA1B: "MSA-1b"}


def checkpoint_args(blob):
    """The hyper-parameters a fair-esm v1 checkpoint carries ({"args": Namespace | dict}) with ONE prefix dropped per
    architecture, as fair-esm's loader drops it before it builds the module (its lambda `pra`, esm/pretrained.py [recalled]):
    `encoder_` for roberta_large and msa_transformer, `decoder_` for protein_bert_base -- `encoder_embed_dim` -> `embed_dim`,
    `decoder_layers` -> `layers`, ...  Keys with the other prefix stay as they are (they are leftovers of the training namespace
    that fair-esm never reads), so a collision such as `encoder_embed_dim` vs `decoder_embed_dim` cannot be resolved by dict
    order.  {} when the file has none."""
    a = blob.get("args") if isinstance(blob, dict) else None
    if a is None:
        return {}
    raw = dict(a) if isinstance(a, dict) else dict(vars(a))
    pre = "decoder_" if raw.get("arch") == "protein_bert_base" else "encoder_"
    out = {}
    for k, v in raw.items():
        if pre in k:
            stripped = "".join(k.split(pre)[1:])
            out[stripped] = v                      # the prefixed form wins over an unprefixed leftover of the same name
    for k, v in raw.items():
        if pre not in k:
            out.setdefault(k, v)
    return out


def config_from_checkpoint(args, state_names, base_cfg, explicit=False):
    """The engine configuration a checkpoint asks for: `base_cfg` (the wrapper's architecture defaults) overridden by the
    checkpoint's own hyper-parameters -- what `esm.pretrained.*` builds the module from in the reference
    (/root/reference/src/pgen/models.py:61-86) -- instead of a fixed dict per wrapper.  Read: arch, embed_dim, layers,
    attention_heads, ffn_embed_dim, max_positions, token_dropout, emb_layer_norm_before, final_bias, embed_positions_msa.
    What fair-esm HONOURS for that architecture and the engine does not implement raises ValueError naming the flag; flags fair-esm
    itself never reads for the architecture (released `args` are the internal training namespace and may carry such leftovers:
    `token_dropout` in an msa_transformer file -- MSATransformer has no token dropout --, `final_bias` outside ESM-1 --
    RobertaLMHead always has a bias --, an `emb_layer_norm_before` value that the tensors contradict -- fair-esm decides it from the
    tensors) are ignored with a warning, as the loader the reference relies on would load the file.  With
    explicit=True (the caller passed `config=`) a disagreement between that config and the file raises instead of silently
    picking one.  NOT validated against the args of the real released .pt files (none are available offline)."""
    import warnings
    cfg = dict(base_cfg)
    arch = args.get("arch")
    if arch is not None:
        if arch not in _ARCH_OF:
            raise ValueError("checkpoint arch %r is not one of the architectures of pgen.models (roberta_large = ESM-1b / ESM-1v, "
                             "protein_bert_base = ESM-1, msa_transformer = ESM-MSA-1b)" % (arch,))
        if _ARCH_OF[arch] != cfg["arch"]:
            raise ValueError("checkpoint arch %r does not match the requested %s engine" % (arch, _ARCH_NAME[cfg["arch"]]))
    take = {}
    for key, name in (("embed_dim", "d_model"), ("layers", "n_layers"), ("attention_heads", "n_heads"), ("ffn_embed_dim", "d_ffn"),
                      ("max_positions", "max_positions")):
        if args.get(key) is not None:
            take[name] = int(args[key])
    esm1b = cfg["arch"] == _lib.PG_ARCH_ESM1B
    if esm1b and "token_dropout" in args:
        take["token_dropout"] = 1 if args["token_dropout"] else 0
    if explicit:
        bad = {k: (cfg[k], v) for k, v in take.items() if cfg[k] != v}
        if bad:
            raise ValueError("config= disagrees with the checkpoint's own hyper-parameters: %s (given, in the file)" % bad)
    cfg.update(take)
    if cfg["n_heads"] * 64 != cfg["d_model"]:
        raise ValueError("checkpoint has %d heads of dimension %g: the engine's attention kernels implement head dimension 64 "
                         "(every model of pgen.models)" % (cfg["n_heads"], cfg["d_model"] / max(1, cfg["n_heads"])))
    has_ln_before = any(n.startswith("emb_layer_norm_before") for n in state_names)
    if cfg["arch"] in (_lib.PG_ARCH_ESM1B, _lib.PG_ARCH_MSA1B):
        # fair-esm decides this flag from the tensors (`has_emb_layer_norm_before`), not from args
        if not has_ln_before:
            raise ValueError("checkpoint was trained without emb_layer_norm_before: the %s engine always applies it (every released "
                             "ESM-1b / ESM-1v / ESM-MSA-1b checkpoint has it)" % _ARCH_NAME[cfg["arch"]])
        if args.get("emb_layer_norm_before") is False:
            warnings.warn("checkpoint args say emb_layer_norm_before=False but the file holds emb_layer_norm_before.*: going by the "
                          "tensors, as fair-esm does")
        if args.get("final_bias") is False:
            warnings.warn("final_bias=False in a %s checkpoint is ignored: only ESM-1's embed_out reads it, RobertaLMHead always has "
                          "a bias" % _ARCH_NAME[cfg["arch"]])
    elif has_ln_before:
        raise ValueError("ESM-1 checkpoint with emb_layer_norm_before tensors: the ESM-1 engine has no embedding LayerNorms")
    elif args.get("emb_layer_norm_before"):
        warnings.warn("checkpoint args say emb_layer_norm_before=True but the file holds no such tensors: going by the tensors, as "
                      "fair-esm does")
    if cfg["arch"] == _lib.PG_ARCH_MSA1B and args.get("embed_positions_msa") is False:
        raise ValueError("embed_positions_msa=False: the MSA engine adds msa_position_embedding (esm_msa1b_t12_100M_UR50S has it)")
    if args.get("token_dropout"):
        if cfg["arch"] == _lib.PG_ARCH_ESM1:
            raise ValueError("token_dropout=True in an ESM-1 checkpoint: fair-esm's ProteinBertModel would apply it, the ESM-1 engine "
                             "does not implement it (none of esm1_t6 / t12 / t34 sets it)")
        if cfg["arch"] == _lib.PG_ARCH_MSA1B:
            warnings.warn("token_dropout=True in an msa_transformer checkpoint is ignored: fair-esm's MSATransformer never reads it")
    n_layers_seen = 1 + max([int(m.group(1)) for m in (re.match(r"layers\.(\d+)\.", n) for n in state_names) if m] or [-1])
    if n_layers_seen and n_layers_seen != cfg["n_layers"]:
        if "n_layers" in take or explicit:
            raise ValueError("checkpoint holds %d layers but its hyper-parameters say %d" % (n_layers_seen, cfg["n_layers"]))
        cfg["n_layers"] = n_layers_seen
    return cfg


def load_fair_esm_checkpoint(path, cfg, return_config=False, explicit_config=False):
    """Read a fair-esm `.pt` file ({"args": ..., "model": state dict}) as the reference does through `esm.pretrained.*`
    (/root/reference/src/pgen/models.py:61,86).  `cfg` gives the architecture (and the defaults of a file without `args`); sizes
    and flags come from the checkpoint's own hyper-parameters (config_from_checkpoint).  return_config=True -> (state dict, cfg)."""
    import torch
    blob = torch.load(path, map_location="cpu", weights_only=False)
    sd = blob["model"] if isinstance(blob, dict) and "model" in blob else blob
    args = checkpoint_args(blob)
    if not args and isinstance(blob, dict) and isinstance(blob.get("cfg"), dict) and isinstance(blob["cfg"].get("model"), dict):
        raise ValueError("checkpoint in fair-esm's v2 layout (ESM-2): not one of the models of pgen.models")
    is_msa = cfg["arch"] == _lib.PG_ARCH_MSA1B
    names = [_swap_row_column(_strip_fair_esm_prefixes(k)) if is_msa else _strip_fair_esm_prefixes(k) for k in sd]
    cfg2 = config_from_checkpoint(args, names, cfg, explicit=explicit_config)
    out = normalise_state_dict(sd, cfg2, fair_esm_layout=True)
    return (out, cfg2) if return_config else out


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
