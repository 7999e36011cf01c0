"""ORACLE (test infrastructure only -- never imported by the product path).

fp32 CPU restatement of the ESM-1 forward pass (`esm1_t6_43M_UR50S`, `esm1_t12_85M_UR50S`, `esm1_t34_670M_UR50S`): the models
behind `pgen.models.ESM6 / ESM12 / ESM34` (/root/reference/src/pgen/models.py:69-82), the only ESM models the reference's own
unit tests load (/root/reference/test/test_esm_sampler.py:10-18, numeric KATs :269-340).

The arithmetic lives in fair-esm (`esm.model.esm1.ProteinBertModel`, model_version "ESM-1", fairseq-style MultiheadAttention
with add_bias_kv=True), which is not installed here; restated from its published code (last paragraph):

  x = sqrt(d) * embed_tokens[tok]                              (embed_scale = sqrt(d); NO token dropout)
  x += sinusoidal(cumsum(tok != pad) * (tok != pad) + pad_idx)  (fairseq SinusoidalPositionalEmbedding: [sin | cos] halves,
                                                                 inv_freq = exp(-i * log(10000) / (d/2 - 1)), pad row zero)
  x *= (tok != pad)                                             (NO emb_layer_norm_before)
  L x { h = LN1(x); q = (W_q h + b_q) * dh^-0.5, k, v; k <- [k ; bias_k], v <- [v ; bias_v] (ONE extra key per sequence, not
        projected, never masked); x += out_proj(softmax(q k^T) v);   x += fc2(gelu(fc1(LN2(x)))) }     (LayerNorm eps 1e-12)
  logits = x @ embed_out^T + embed_out_bias                     (untied; NO emb_layer_norm_after, NO lm_head.dense / layer_norm)

Vocabulary (alphabet "ESM-1"): <null_0> <pad> <eos> <unk> + 27 residue symbols + <null_1> + <cls>=32 <mask>=33 <sep>=34; V = 35;
prepend_bos (<cls>), no eos.

PARITY PINNING: **parity unpinned** against the reference (its KATs need the pretrained 43 M checkpoint; offline).  The
attention block with bias_k / bias_v is cross-checked against torch.nn.MultiheadAttention(add_bias_kv=True), an independent
implementation of the same fairseq-derived semantics (tests/test_oracle_esm1.py).
"""
import numpy as np

from .esm_forward import F32, gelu, layer_norm, linear, softmax_lastdim


class Esm1Config:
    def __init__(self, vocab=35, d_model=768, n_layers=6, n_heads=12, d_ffn=3072, max_pos=1024, pad_idx=1, mask_idx=33,
                 cls_idx=32, eos_idx=2, final_bias=True):
        self.vocab, self.d_model, self.n_layers, self.n_heads, self.d_ffn = vocab, d_model, n_layers, n_heads, d_ffn
        self.max_pos, self.pad_idx, self.mask_idx, self.cls_idx, self.eos_idx = max_pos, pad_idx, mask_idx, cls_idx, eos_idx
        self.final_bias = final_bias


def sinusoidal_table(n_rows, d, pad_idx):
    """fairseq / fair-esm SinusoidalPositionalEmbedding.get_embedding, computed in float32 as torch does."""
    half = d // 2
    step = F32(np.log(10000.0) / (half - 1))
    inv = np.exp(np.arange(half, dtype=F32) * -step).astype(F32)
    ang = (np.arange(n_rows, dtype=F32)[:, None] * inv[None, :]).astype(F32)
    emb = np.concatenate([np.sin(ang), np.cos(ang)], axis=1).astype(F32)
    if d % 2:
        emb = np.concatenate([emb, np.zeros((n_rows, 1), F32)], axis=1)
    emb[pad_idx] = 0
    return emb


def esm1_embed(w, cfg, tokens):
    tokens = np.asarray(tokens)
    pad = tokens == cfg.pad_idx
    x = (F32(np.sqrt(F32(cfg.d_model))) * w["embed_tokens.weight"][tokens]).astype(F32)
    nonpad = (~pad).astype(np.int64)
    pos = np.cumsum(nonpad, axis=1) * nonpad + cfg.pad_idx
    x = x + sinusoidal_table(cfg.pad_idx + 1 + tokens.shape[1], cfg.d_model, cfg.pad_idx)[pos]
    return np.where(pad[..., None], F32(0), x).astype(F32), pad


def mha_bias_kv(w, prefix, cfg, h, pad):
    B, T, d = h.shape
    H = cfg.n_heads
    dh = d // H
    q = linear(h, w[prefix + "q_proj.weight"], w[prefix + "q_proj.bias"]) * F32(dh ** -0.5)
    k = linear(h, w[prefix + "k_proj.weight"], w[prefix + "k_proj.bias"])
    v = linear(h, w[prefix + "v_proj.weight"], w[prefix + "v_proj.bias"])
    k = np.concatenate([k, np.broadcast_to(w[prefix + "bias_k"].reshape(1, 1, d), (B, 1, d))], axis=1)      # [B, T+1, d]
    v = np.concatenate([v, np.broadcast_to(w[prefix + "bias_v"].reshape(1, 1, d), (B, 1, d))], axis=1)
    q = q.reshape(B, T, H, dh).transpose(0, 2, 1, 3)
    k = k.reshape(B, T + 1, H, dh).transpose(0, 2, 1, 3)
    v = v.reshape(B, T + 1, H, dh).transpose(0, 2, 1, 3)
    a = (q @ k.transpose(0, 1, 3, 2)).astype(F32)                        # [B,H,T,T+1]
    if pad.any():
        kpm = np.concatenate([pad, np.zeros((B, 1), bool)], axis=1)       # the extra key is never padding
        a = np.where(kpm[:, None, None, :], F32(-np.inf), a)
    p = softmax_lastdim(a)
    ctx = (p @ v).transpose(0, 2, 1, 3).reshape(B, T, d).astype(F32)
    return linear(ctx, w[prefix + "out_proj.weight"], w[prefix + "out_proj.bias"])


def esm1_trunk(w, cfg, tokens):
    x, pad = esm1_embed(w, cfg, tokens)
    eps = 1e-12
    for i in range(cfg.n_layers):
        p = "layers.%d." % i
        h = layer_norm(x, w[p + "self_attn_layer_norm.weight"], w[p + "self_attn_layer_norm.bias"], eps)
        x = x + mha_bias_kv(w, p + "self_attn.", cfg, h, pad)
        h = layer_norm(x, w[p + "final_layer_norm.weight"], w[p + "final_layer_norm.bias"], eps)
        h = gelu(linear(h, w[p + "fc1.weight"], w[p + "fc1.bias"]))
        x = (x + linear(h, w[p + "fc2.weight"], w[p + "fc2.bias"])).astype(F32)
    return x


def esm1_head(w, x):
    return (x @ w["embed_out.weight"].T + w["embed_out.bias"]).astype(F32)


def esm1_forward(w, cfg, tokens):
    """tokens int [B,T] -> logits fp32 [B,T,35].  Keys: the engine's names -- `embed_out.weight` / `embed_out.bias` for fair-esm's
    `embed_out` / `embed_out_bias`, `layers.i.self_attn.bias_k` / `bias_v` flattened to [d]."""
    return esm1_head(w, esm1_trunk(w, cfg, tokens))


def synthetic_esm1_weights(cfg, seed=0, std=0.02, embed_std=None, ln_jitter=0.0):
    rng = np.random.default_rng(seed)
    d, f, V = cfg.d_model, cfg.d_ffn, cfg.vocab
    es = std if embed_std is None else embed_std

    def n(*shape, s=std):
        return (rng.standard_normal(shape, dtype=np.float32) * np.float32(s)).astype(F32)

    def ln(prefix, w):
        w[prefix + ".weight"] = (1.0 + ln_jitter * rng.standard_normal(d, dtype=np.float32)).astype(F32)
        w[prefix + ".bias"] = (ln_jitter * rng.standard_normal(d, dtype=np.float32)).astype(F32)

    w = {"embed_tokens.weight": n(V, d, s=es)}
    for i in range(cfg.n_layers):
        p = "layers.%d." % i
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            w[p + "self_attn." + nm + ".weight"] = n(d, d)
            w[p + "self_attn." + nm + ".bias"] = n(d)
        w[p + "self_attn.bias_k"] = n(d, s=0.3)
        w[p + "self_attn.bias_v"] = n(d, s=0.3)
        ln(p + "self_attn_layer_norm", w)
        w[p + "fc1.weight"] = n(f, d)
        w[p + "fc1.bias"] = n(f)
        w[p + "fc2.weight"] = n(d, f)
        w[p + "fc2.bias"] = n(d)
        ln(p + "final_layer_norm", w)
    w["embed_out.weight"] = n(V, d, s=es)
    w["embed_out.bias"] = n(V)
    return w
