"""ORACLE (test infrastructure only -- never imported by the product path).

fp32 CPU restatement of the ESM-1b forward pass the reference calls as
`self.model.model(batch)["logits"]` (/root/reference/src/pgen/esm_sampler.py:223).

The arithmetic is NOT in /root/reference: it lives in the third-party package
`fair-esm` (module `esm`, unpinned git HEAD -- /root/reference/conda_env.yml:21,
README.md:34; call sites models.py:61, esm_sampler.py:223,340,355), which is not
installed here.  This file restates its published algorithm
(`esm.model.esm1.ProteinBertModel.forward`, "ESM-1b" architecture; SURVEY.md
Appendix A.2):

  x = embed_tokens[tok]                                   (embed_scale = 1)
  token dropout: x[tok == mask] = 0;  x *= (1 - 0.15*0.8) / (1 - n_mask_b / src_len_b)
  x += embed_positions[cumsum(tok != pad) * (tok != pad) + pad_idx]
  x = LN_before(x);  x *= (tok != pad)
  33 x { x += out_proj(MHA(LN1(x)));  x += fc2(gelu(fc1(LN2(x)))) }   (pre-LN, erf GELU)
  x = LN_after(x)
  logits = LN(gelu(dense(x))) @ embed_tokens^T + bias      (tied decoder)

PARITY PINNING: the reference's own numeric KATs for this boundary
(/root/reference/test/test_esm_sampler.py:269-340) need the pretrained esm1_t6
checkpoint, which is unavailable offline => against the reference itself the
logits are "parity unpinned".  The restatement is instead corroborated against an
independent implementation of the same architecture, HuggingFace
`transformers.EsmForMaskedLM` (tests/golden/make_golden.py -> esm_hf_*.npz).

Weights are a dict keyed by fair-esm state-dict names.
"""
import numpy as np
from scipy.special import erf

F32 = np.float32


def layer_norm(x, w, b, eps=1e-5):
    x = x.astype(F32)
    mu = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(eps)) * w + b).astype(F32)


def _gelu_block(x):
    return (F32(0.5) * x * (F32(1.0) + erf(x * F32(0.7071067811865476)))).astype(F32)


def gelu(x):
    """exact (erf) GELU, elementwise; big arrays in row blocks on a thread pool (scipy's erf is a single-threaded ufunc that releases
    the GIL: 30 % of a full-size forward otherwise) -- the same value per element either way."""
    x = np.ascontiguousarray(x, dtype=F32)
    if x.size < (1 << 22) or x.ndim < 2:
        return _gelu_block(x)
    import os
    from concurrent.futures import ThreadPoolExecutor
    flat = x.reshape(-1, x.shape[-1])
    n = min(os.cpu_count() or 1, 64, flat.shape[0])
    out = np.empty_like(flat)
    bounds = np.linspace(0, flat.shape[0], n + 1).astype(int)

    def run(i):
        out[bounds[i]:bounds[i + 1]] = _gelu_block(flat[bounds[i]:bounds[i + 1]])
    with ThreadPoolExecutor(n) as ex:
        list(ex.map(run, range(n)))
    return out.reshape(x.shape)


def linear(x, w, b):
    return (x @ w.T + b).astype(F32)


def softmax_lastdim(a):
    a = a - a.max(axis=-1, keepdims=True)
    e = np.exp(a, dtype=F32)
    return (e / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)


class EsmConfig:
    def __init__(self, vocab=33, d_model=1280, n_layers=33, n_heads=20, d_ffn=5120, max_pos=1024,
                 pad_idx=1, mask_idx=32, cls_idx=0, eos_idx=2, token_dropout=True):
        self.vocab, self.d_model, self.n_layers, self.n_heads, self.d_ffn = vocab, d_model, n_layers, n_heads, d_ffn
        self.max_pos, self.pad_idx, self.mask_idx, self.cls_idx, self.eos_idx = max_pos, pad_idx, mask_idx, cls_idx, eos_idx
        self.token_dropout = token_dropout


def esm1b_embed(w, cfg, tokens):
    tokens = np.asarray(tokens)
    B, T = tokens.shape
    pad = tokens == cfg.pad_idx
    x = w["embed_tokens.weight"][tokens].astype(F32)                    # [B,T,d]
    if cfg.token_dropout:
        is_mask = tokens == cfg.mask_idx
        x = np.where(is_mask[..., None], F32(0), x)
        src_len = (~pad).sum(axis=1).astype(F32)
        ratio = is_mask.sum(axis=1).astype(F32) / src_len
        scale = (F32(1 - 0.15 * 0.8) / (F32(1) - ratio)).astype(F32)
        x = (x * scale[:, None, None]).astype(F32)
    nonpad = (~pad).astype(np.int64)
    pos = np.cumsum(nonpad, axis=1) * nonpad + cfg.pad_idx
    x = x + w["embed_positions.weight"][pos]
    x = layer_norm(x, w["emb_layer_norm_before.weight"], w["emb_layer_norm_before.bias"])
    x = np.where(pad[..., None], F32(0), x).astype(F32)
    return x, pad


def mha(w, prefix, cfg, h, pad):
    B, T, d = h.shape
    H = cfg.n_heads
    dh = d // H
    q = linear(h, w[prefix + "q_proj.weight"], w[prefix + "q_proj.bias"]) * F32(dh ** -0.5)
    k = linear(h, w[prefix + "k_proj.weight"], w[prefix + "k_proj.bias"])
    v = linear(h, w[prefix + "v_proj.weight"], w[prefix + "v_proj.bias"])
    q = q.reshape(B, T, H, dh).transpose(0, 2, 1, 3)
    k = k.reshape(B, T, H, dh).transpose(0, 2, 1, 3)
    v = v.reshape(B, T, H, dh).transpose(0, 2, 1, 3)
    a = (q @ k.transpose(0, 1, 3, 2)).astype(F32)                        # [B,H,T,T]
    if pad.any():
        a = np.where(pad[:, None, None, :], F32(-np.inf), a)
    p = softmax_lastdim(a)
    ctx = (p @ v).transpose(0, 2, 1, 3).reshape(B, T, d).astype(F32)
    return linear(ctx, w[prefix + "out_proj.weight"], w[prefix + "out_proj.bias"])


def esm1b_trunk(w, cfg, tokens, return_layers=False):
    """Everything up to and including emb_layer_norm_after: returns x [B,T,d] fp32."""
    x, pad = esm1b_embed(w, cfg, tokens)
    layers = [x.copy()] if return_layers else None
    for i in range(cfg.n_layers):
        p = "layers.%d." % i
        h = layer_norm(x, w[p + "self_attn_layer_norm.weight"], w[p + "self_attn_layer_norm.bias"])
        x = x + mha(w, p + "self_attn.", cfg, h, pad)
        h = layer_norm(x, w[p + "final_layer_norm.weight"], w[p + "final_layer_norm.bias"])
        h = gelu(linear(h, w[p + "fc1.weight"], w[p + "fc1.bias"]))
        x = (x + linear(h, w[p + "fc2.weight"], w[p + "fc2.bias"])).astype(F32)
        if return_layers:
            layers.append(x.copy())
    x = layer_norm(x, w["emb_layer_norm_after.weight"], w["emb_layer_norm_after.bias"])
    return (x, layers) if return_layers else x


def lm_head(w, x):
    h = gelu(linear(x, w["lm_head.dense.weight"], w["lm_head.dense.bias"]))
    h = layer_norm(h, w["lm_head.layer_norm.weight"], w["lm_head.layer_norm.bias"])
    return (h @ w["embed_tokens.weight"].T + w["lm_head.bias"]).astype(F32)


def esm1b_forward(w, cfg, tokens):
    """tokens int [B,T] -> logits fp32 [B,T,V]."""
    return lm_head(w, esm1b_trunk(w, cfg, tokens))


def synthetic_esm_weights(cfg, seed=0, std=0.02, embed_std=None, ln_jitter=0.0):
    """Seeded synthetic weights with fair-esm key names (no checkpoints offline).

    N(0, std^2) linears/embeddings, LayerNorm gamma=1 (+jitter) beta=0 (+jitter), small biases.
    numpy's PCG64 stream is stable across versions, so tests regenerate instead of
    storing MB-scale fixtures.
    """
    rng = np.random.default_rng(seed)
    d, f, V = cfg.d_model, cfg.d_ffn, cfg.vocab
    es = std if embed_std is None else embed_std

    def n(*shape, s=std):
        return (rng.standard_normal(shape, dtype=np.float32) * np.float32(s)).astype(F32)

    def ln(prefix, w):
        w[prefix + ".weight"] = (1.0 + ln_jitter * rng.standard_normal(d, dtype=np.float32)).astype(F32)
        w[prefix + ".bias"] = (ln_jitter * rng.standard_normal(d, dtype=np.float32)).astype(F32)

    w = {}
    w["embed_tokens.weight"] = n(V, d, s=es)
    w["embed_positions.weight"] = n(cfg.max_pos + cfg.pad_idx + 1, d, s=es)
    ln("emb_layer_norm_before", w)
    for i in range(cfg.n_layers):
        p = "layers.%d." % i
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            w[p + "self_attn." + nm + ".weight"] = n(d, d)
            w[p + "self_attn." + nm + ".bias"] = n(d)
        ln(p + "self_attn_layer_norm", w)
        w[p + "fc1.weight"] = n(f, d)
        w[p + "fc1.bias"] = n(f)
        w[p + "fc2.weight"] = n(d, f)
        w[p + "fc2.bias"] = n(d)
        ln(p + "final_layer_norm", w)
    ln("emb_layer_norm_after", w)
    w["lm_head.dense.weight"] = n(d, d)
    w["lm_head.dense.bias"] = n(d)
    ln("lm_head.layer_norm", w)
    w["lm_head.bias"] = n(V)
    return w
